"""Cross-agent keyframe descriptor exchange: one agent per GPU, RCCL all-gather over xGMI.

In the reference the server asks every other agent's BoW inverted index for loop/merge candidates of each new
keyframe (AgentMediator::CheckOverlapCandidates, code/src/AgentMediator.cc:140-202).  With one agent per GPU the
same question is answered without a server: every `tick`, each rank contributes ONE fixed-capacity slot holding
its newest keyframe's descriptors (header + slot_keypoints x 32 B; 32-64 KB), a single `all_gather_into_tensor`
(backend "nccl" == RCCL on ROCm) delivers all slots to all ranks, and each rank brute-force matches its own slot
against every peer slot with the Hamming top-2 kernel (so_hamming_top2_device).  The payload is latency-bound
(tens of microseconds), far below the per-link xGMI limit, so one collective per tick on padded slots is the
right shape; there is no per-frame collective.

The exchange itself only needs torch.distributed, so it is covered by world_size-2 gloo tests on CPU; the
matching step needs the GPU library.
"""
import numpy as np
import torch
import torch.distributed as dist

HEADER_ROWS = 1  # row 0 of a slot: int32 keypoint count, int32 rank, int64 checksum of the descriptor bytes


def slot_checksum(desc):
    """Order-sensitive 63-bit checksum of a descriptor block (used to verify gathered payloads)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1)
    if d.size == 0:
        return 0
    w = (np.arange(d.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return int((d.astype(np.uint64) * w).sum() % np.uint64((1 << 61) - 1))


class KeyframeExchange:
    def __init__(self, slot_keypoints=2024, device=None, group=None):
        if not dist.is_initialized():
            raise RuntimeError("KeyframeExchange needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.slot_keypoints = int(slot_keypoints)
        on_gpu = dist.get_backend(group) == "nccl"
        self.device = torch.device("cuda", device if device is not None else torch.cuda.current_device()) \
            if on_gpu else torch.device("cpu")
        rows = HEADER_ROWS + self.slot_keypoints
        self.slot = torch.zeros((rows, 32), dtype=torch.uint8, device=self.device)
        self.gathered = torch.zeros((self.world, rows, 32), dtype=torch.uint8, device=self.device)
        self._host_slot = torch.zeros((rows, 32), dtype=torch.uint8).pin_memory() if on_gpu \
            else torch.zeros((rows, 32), dtype=torch.uint8)

    def exchange(self, desc):
        """All-gather one keyframe's descriptors.  Returns (gathered [world, rows, 32] tensor, counts, checksums)."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = min(len(desc), self.slot_keypoints)
        hs = self._host_slot.numpy()
        hs[0, :] = 0
        hs[0, 0:4] = np.frombuffer(np.int32(n).tobytes(), np.uint8)
        hs[0, 4:8] = np.frombuffer(np.int32(self.rank).tobytes(), np.uint8)
        hs[0, 8:16] = np.frombuffer(np.int64(slot_checksum(desc[:n])).tobytes(), np.uint8)
        hs[HEADER_ROWS:HEADER_ROWS + n] = desc[:n]
        self.slot.copy_(self._host_slot, non_blocking=True)
        if self.device.type == "cuda":
            dist.all_gather_into_tensor(self.gathered, self.slot, group=self.group)  # one RCCL all-gather
        else:  # gloo (CPU tests): list form over views of the same buffer
            dist.all_gather(list(self.gathered.unbind(0)), self.slot, group=self.group)
        hdr = self.gathered[:, 0, :16].cpu().numpy()  # synchronises with the collective
        counts = hdr[:, 0:4].copy().view(np.int32).reshape(-1)
        ranks = hdr[:, 4:8].copy().view(np.int32).reshape(-1)
        sums = hdr[:, 8:16].copy().view(np.int64).reshape(-1)
        assert np.array_equal(ranks, np.arange(self.world)), "all-gather slot order"
        return self.gathered, counts, sums

    def exchange_and_match(self, desc, matcher, max_dist=50, ratio=0.75):
        """Exchange, then match this rank's keyframe against every peer's (GPU).  Returns
        {peer: number of mutual-ratio-test candidates} — what the host merger would be told."""
        gathered, counts, _ = self.exchange(desc)
        torch.cuda.synchronize(self.device)
        row_bytes = gathered.shape[1] * 32
        base = gathered.data_ptr()
        mine = base + self.rank * row_bytes + HEADER_ROWS * 32
        out = {}
        for peer in range(self.world):
            if peer == self.rank or counts[peer] == 0 or counts[self.rank] == 0:
                continue
            theirs = base + peer * row_bytes + HEADER_ROWS * 32
            bi, bd, sd = matcher.hamming_top2_device(mine, int(counts[self.rank]), theirs, int(counts[peer]))
            out[peer] = int(((bd <= max_dist) & (bd < ratio * sd)).sum())
        return out
