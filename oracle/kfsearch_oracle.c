/* kfsearch_oracle.c — CPU restatement of the cross-agent keyframe candidate search.  TEST INFRASTRUCTURE ONLY: imported
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by swarmmap_amd/.
 *
 * Reference: AgentMediator::CheckOverlapCandidates (code/src/AgentMediator.cc:140-202) looks every new keyframe up in
 * every OTHER agent's whole keyframe database (KeyFrameDatabase::DetectLoopCandidates, BoW scores), keeps the candidates
 * of another origin map (:186) and hands them to AgentMediator::GetSim3 (:204-262), which runs
 * ORBmatcher(0.75, true).SearchByBoW(pCurrentKF, pKF, ...) per candidate and drops those with fewer than 20 pairs
 * (:259).  The vocabulary (ORBvoc.bin) is not part of the checkout and the north star replaces the BoW
 * query by brute-force Hamming matching with the ratio test, so the detection score is restated here as
 *     votes(k) = #{ i1 bound in the query : best(i1, k) < TH_LOW and (float)best < ratio * (float)second }
 * with best / second found by the scan of code/src/ORBmatcher.cc:524-549 over keyframe k's bound keypoints (nothing is
 * "taken" yet) and the acceptance test of :550-551.  Phase 2 is orc_search_by_bow(variant 1) - the restatement of
 * :481-597 in matcher_oracle.c - with all features in one vocabulary node; the composition lives in oracle_py.kf_search.
 * parity unpinned: like the rest of the oracle, nothing of the reference compiles here. */
#include <stddef.h>
#include <stdint.h>

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);

int orc_kf_votes(int32_t n1, const uint8_t* desc1, const uint8_t* valid1, int32_t n2, const uint8_t* desc2,
                 const uint8_t* valid2, int32_t th_low, float nn_ratio) {
    int votes = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        if (!valid1[i1]) continue; /* :517-521 */
        int bestDist1 = 256, bestDist2 = 256;
        for (int i2 = 0; i2 < n2; i2++) {
            if (!valid2[i2]) continue; /* :535-541 */
            const int dist = orc_descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
            if (dist < bestDist1) { /* :543-549 */
                bestDist2 = bestDist1;
                bestDist1 = dist;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist1 < th_low && (float)bestDist1 < nn_ratio * (float)bestDist2) votes++; /* :550-551 */
    }
    return votes;
}
