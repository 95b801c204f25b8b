// quadtree.cpp — see quadtree.h.  Follows code/src/ORBextractor.cc:407-689.
#include "quadtree.h"

#include <algorithm>
#include <cmath>

namespace so {

int KeypointQuadtree::new_node(int x0, int y0, int x1, int y1, int cap) {
    Node nd;
    nd.x0 = x0; nd.y0 = y0; nd.x1 = x1; nd.y1 = y1;
    nd.off = pool_used_;
    nd.n = 0;
    nd.prev = nd.next = -1;
    nd.leaf = false;
    pool_used_ += cap;
    if ((size_t)pool_used_ > pool_.size()) pool_.resize((size_t)pool_used_ * 2 + 1024);  // grows only in the first frames
    nodes_.push_back(nd);
    return (int)nodes_.size() - 1;
}

void KeypointQuadtree::push_front(int id) {
    nodes_[id].prev = -1;
    nodes_[id].next = head_;
    if (head_ >= 0) nodes_[head_].prev = id;
    head_ = id;
    if (tail_ < 0) tail_ = id;
    ++size_;
}

void KeypointQuadtree::push_back(int id) {
    nodes_[id].next = -1;
    nodes_[id].prev = tail_;
    if (tail_ >= 0) nodes_[tail_].next = id;
    tail_ = id;
    if (head_ < 0) head_ = id;
    ++size_;
}

int KeypointQuadtree::erase(int id) {
    const int p = nodes_[id].prev, n = nodes_[id].next;
    if (p >= 0) nodes_[p].next = n; else head_ = n;
    if (n >= 0) nodes_[n].prev = p; else tail_ = p;
    --size_;
    return n;
}

// ExtractorNode::DivideNode (:407-463)
void KeypointQuadtree::split(int id, const Candidate* c, int child[4]) {
    const Node P = nodes_[id];
    const int half_x = (int)std::ceil((float)(P.x1 - P.x0) / 2);
    const int half_y = (int)std::ceil((float)(P.y1 - P.y0) / 2);
    const int xm = P.x0 + half_x, ym = P.y0 + half_y;
    // two passes: count the quadrant populations, then place -> the children use exactly P.n pool slots
    if ((int)quad_.size() < P.n) quad_.resize((size_t)P.n * 2 + 64);
    int cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < P.n; i++) {
        const int ci = pool_[P.off + i];
        const int q = (c[ci].x < xm ? 0 : 1) + (c[ci].y < ym ? 0 : 2);
        quad_[(size_t)i] = (uint8_t)q;
        cnt[q]++;
    }
    child[0] = new_node(P.x0, P.y0, xm, ym, cnt[0]);
    child[1] = new_node(xm, P.y0, P.x1, ym, cnt[1]);
    child[2] = new_node(P.x0, ym, xm, P.y1, cnt[2]);
    child[3] = new_node(xm, ym, P.x1, P.y1, cnt[3]);
    for (int i = 0; i < P.n; i++) {
        Node& d = nodes_[child[quad_[(size_t)i]]];
        pool_[d.off + d.n++] = pool_[P.off + i];
    }
    for (int k = 0; k < 4; k++)
        if (nodes_[child[k]].n == 1) nodes_[child[k]].leaf = true;
}

void KeypointQuadtree::link_children(const int child[4], bool count_expand, int& n_to_expand) {
    for (int k = 0; k < 4; k++) {
        const int n = nodes_[child[k]].n;
        if (n > 0) {
            push_front(child[k]);
            if (n > 1) {
                if (count_expand) ++n_to_expand;
                expand_.emplace_back(n, child[k]);
            }
        }
    }
}

int KeypointQuadtree::distribute(const Candidate* c, int n, int roi_w, int roi_h, int N, std::vector<int>& out) {
    out.clear();
    if (n <= 0) return 0;
    nodes_.clear();
    if (nodes_.capacity() < 8192) nodes_.reserve(8192);
    pool_used_ = 0;
    head_ = tail_ = -1;
    size_ = 0;

    // :468-496 root nodes
    int n_ini = (int)std::round((float)roi_w / (float)roi_h);
    if (n_ini < 1) n_ini = 1;  // reference divides by zero here for very tall images
    const float hx = (float)roi_w / (float)n_ini;
    roots_.resize((size_t)n_ini);
    for (int i = 0; i < n_ini; i++) {
        roots_[(size_t)i] = new_node((int)(hx * (float)i), 0, (int)(hx * (float)(i + 1)), roi_h, n);
        push_back(roots_[(size_t)i]);
    }
    for (int i = 0; i < n; i++) {
        int r = (int)((float)c[i].x / hx);
        if (r >= n_ini) r = n_ini - 1;
        Node& d = nodes_[roots_[(size_t)r]];
        pool_[d.off + d.n++] = i;
    }
    for (int it = head_; it >= 0;) {  // :498-511
        if (nodes_[it].n == 1) {
            nodes_[it].leaf = true;
            it = nodes_[it].next;
        } else if (nodes_[it].n == 0) {
            it = erase(it);
        } else {
            it = nodes_[it].next;
        }
    }

    bool finish = false;
    while (!finish) {
        const int prev_size = size_;
        int n_to_expand = 0;
        expand_.clear();
        for (int it = head_; it >= 0;) {  // sweep: children go to the front, so they are not revisited
            if (nodes_[it].leaf) {
                it = nodes_[it].next;
                continue;
            }
            int child[4];
            split(it, c, child);
            link_children(child, true, n_to_expand);
            it = erase(it);
        }
        if (size_ >= N || size_ == prev_size) {
            finish = true;
        } else if (size_ + n_to_expand * 3 > N) {  // careful phase :599-664
            while (!finish) {
                const int prev2 = size_;
                prev_expand_.swap(expand_);
                expand_.clear();
                std::sort(prev_expand_.begin(), prev_expand_.end());
                for (int j = (int)prev_expand_.size() - 1; j >= 0; j--) {
                    const int id = prev_expand_[(size_t)j].second;
                    int child[4], dummy = 0;
                    split(id, c, child);
                    link_children(child, false, dummy);
                    erase(id);
                    if (size_ >= N) break;
                }
                if (size_ >= N || size_ == prev2) finish = true;
            }
        }
    }
    // :667-686 best response per node (first maximum)
    for (int it = head_; it >= 0; it = nodes_[it].next) {
        const Node& nd = nodes_[it];
        int best = pool_[nd.off];
        int best_score = c[best].score;
        for (int k = 1; k < nd.n; k++) {
            const int ci = pool_[nd.off + k];
            if ((int)c[ci].score > best_score) {
                best = ci;
                best_score = c[ci].score;
            }
        }
        out.push_back(best);
    }
    return (int)out.size();
}

}  // namespace so
