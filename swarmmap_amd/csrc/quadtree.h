// quadtree.h — host-side keypoint distribution (ORBextractor::DistributeOctTree,
// code/src/ORBextractor.cc:407-689) on flat candidate records.
//
// Same splitting semantics as the reference (ceil-half splits, children pushed to the FRONT of the node
// list in the order n1..n4, sweep until >= N nodes or no growth, then "careful" largest-first splitting,
// first-maximum response per node), but on an index arena instead of std::list<ExtractorNode> holding
// cv::KeyPoint copies.  The reference's one allocator-dependent choice — sorting (population, node pointer)
// pairs, :610 — is replaced by (population, creation sequence): among equal populations the most recently
// created node is split first.
#pragma once
#include <cstdint>
#include <vector>

#include "orb_device.h"

namespace so {

class KeypointQuadtree {
public:
    // cands: n records of one level (ROI-relative coords). Writes the indices (into cands) of the kept
    // keypoints in output order; returns how many (<= N + 2).
    int distribute(const Candidate* cands, int n, int roi_w, int roi_h, int N, std::vector<int>& out);

private:
    struct Node {
        int x0, y0, x1, y1;
        int off, n;  // key slice in pool_
        int prev, next;
        bool leaf;  // bNoMore
    };
    std::vector<Node> nodes_;
    std::vector<int> pool_;   // key slices; high-water sized, reused frame after frame
    int pool_used_ = 0;
    std::vector<uint8_t> quad_;  // scratch: quadrant of each key of the node being split
    std::vector<int> roots_;
    std::vector<std::pair<int, int>> expand_, prev_expand_;  // (population, creation seq == node id)
    int head_ = -1, tail_ = -1, size_ = 0;

    int new_node(int x0, int y0, int x1, int y1, int cap);
    void push_front(int id);
    void push_back(int id);
    int erase(int id);
    void split(int id, const Candidate* c, int child[4]);
    void link_children(const int child[4], bool count_expand, int& n_to_expand);
};

}  // namespace so
