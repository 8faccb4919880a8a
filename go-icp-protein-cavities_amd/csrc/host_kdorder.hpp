// host_kdorder.hpp -- tie-breaking companion of the device 1-NN search (k_icp_nn).
//
// The reference finds ICP correspondences with a kd-tree (vendored nanoflann 1.1.6, leaf size 10, built by ICP3D::Build,
// reference jly_icp3d.hpp:141-169). Its search is exact, but a candidate replaces the incumbent only when it is STRICTLY
// closer (nanoflann.hpp:95-116, 1147), so when several model points sit at exactly the same float distance from a query the
// winner is the one the traversal meets first -- a function of the tree's layout (divideTree / middleSplit_ / planeSplit,
// nanoflann.hpp:927-1110) and of the near-child-first descent (nanoflann.hpp:1137-1184), not of the point index.
// The device search is exhaustive and reports how many model points share the minimal distance; only for queries where that
// count exceeds one does the host walk this restated tree (built lazily, once per registration) to name the reference's choice.
#ifndef GOICP_HOST_KDORDER_HPP
#define GOICP_HOST_KDORDER_HPP
#include <cfloat>
#include <cstddef>
#include <utility>
#include <vector>

namespace goicp {

class KdOrder {
public:
    KdOrder() {}
    bool built() const { return built_; }
    void clear() { built_ = false; order_.clear(); nodes_.clear(); xyz_.clear(); }

    // xyz: n model points, 3 floats each (copied)
    void build(const float* xyz, int n) {
        clear();
        xyz_.assign(xyz, xyz + (size_t)3 * n);
        order_.resize((size_t)n);
        for (int i = 0; i < n; i++) order_[(size_t)i] = i;
        built_ = true; root_ = -1;
        if (n <= 0) return;
        for (int a = 0; a < 3; a++) { box_[a].lo = box_[a].hi = at(0, a); }
        for (int k = 1; k < n; k++) for (int a = 0; a < 3; a++) { const float v = at(k, a); if (v < box_[a].lo) box_[a].lo = v; if (v > box_[a].hi) box_[a].hi = v; }
        root_ = split(0, (size_t)n, box_);
    }

    // the model index the reference's knnSearch(q, 1) returns; *dist receives the squared distance it reports
    int nearest(const float q[3], float* dist = nullptr) const {
        Best b; b.d = FLT_MAX; b.id = 0;
        float side[3] = {0.f, 0.f, 0.f}, acc = 0.f;
        for (int a = 0; a < 3; a++) {                               // computeInitialDistances, nanoflann.hpp:1113-1131
            if (q[a] < box_[a].lo) { side[a] = (q[a] - box_[a].lo) * (q[a] - box_[a].lo); acc += side[a]; }
            if (q[a] > box_[a].hi) { side[a] = (q[a] - box_[a].hi) * (q[a] - box_[a].hi); acc += side[a]; }
        }
        if (root_ >= 0) descend(root_, q, acc, side, b);
        if (dist) *dist = b.d;
        return b.id;
    }

private:
    struct Span { float lo, hi; };
    struct Node { int near_lo = -1, near_hi = -1; size_t first = 0, last = 0; int axis = 0; float lo_edge = 0, hi_edge = 0; };
    struct Best { float d; int id; };
    static constexpr size_t kLeaf = 10;

    float at(int idx, int a) const { return xyz_[(size_t)3 * idx + a]; }

    void extent(const int* ind, size_t count, int a, float& mn, float& mx) const {
        mn = mx = at(ind[0], a);
        for (size_t i = 1; i < count; i++) { const float v = at(ind[i], a); if (v < mn) mn = v; if (v > mx) mx = v; }
    }

    // planeSplit, nanoflann.hpp:1084-1110: [0,l1) < cut, [l1,l2) == cut, [l2,count) > cut, with its swap order
    void partition(int* ind, size_t count, int a, float cut, size_t& l1, size_t& l2) const {
        size_t i = 0, j = count - 1;
        for (;;) {
            while (i <= j && at(ind[i], a) < cut) ++i;
            while (j && i <= j && at(ind[j], a) >= cut) --j;
            if (i > j || !j) break;
            std::swap(ind[i], ind[j]); ++i; --j;
        }
        l1 = i; j = count - 1;
        for (;;) {
            while (i <= j && at(ind[i], a) <= cut) ++i;
            while (j && i <= j && at(ind[j], a) > cut) --j;
            if (i > j || !j) break;
            std::swap(ind[i], ind[j]); ++i; --j;
        }
        l2 = i;
    }

    // divideTree + middleSplit_, nanoflann.hpp:927-973,1033-1074; `box` comes in as the parent's estimate and leaves as this subtree's bounds
    int split(size_t first, size_t last, Span box[3]) {
        const int me = (int)nodes_.size();
        nodes_.emplace_back();
        if (last - first <= kLeaf) {
            nodes_[(size_t)me].first = first; nodes_[(size_t)me].last = last;
            for (int a = 0; a < 3; a++) box[a].lo = box[a].hi = at(order_[first], a);
            for (size_t k = first + 1; k < last; k++) for (int a = 0; a < 3; a++) {
                const float v = at(order_[k], a);
                if (box[a].lo > v) box[a].lo = v;
                if (box[a].hi < v) box[a].hi = v;
            }
            return me;
        }
        int* ind = order_.data() + first; const size_t count = last - first;
        const float eps = 0.00001f;
        float widest = box[0].hi - box[0].lo;
        for (int a = 1; a < 3; a++) { const float w = box[a].hi - box[a].lo; if (w > widest) widest = w; }
        float best_spread = -1; int axis = 0; float mn, mx;
        for (int a = 0; a < 3; a++) {
            const float w = box[a].hi - box[a].lo;
            if (w > (1 - eps) * widest) {
                extent(ind, count, axis, mn, mx);                   // the axis picked so far, as the reference does (nanoflann.hpp:1047)
                const float spread = mx - mn;
                if (spread > best_spread) { axis = a; best_spread = spread; }
            }
        }
        const float mid = (box[axis].lo + box[axis].hi) / 2;
        extent(ind, count, axis, mn, mx);
        const float cut = mid < mn ? mn : (mid > mx ? mx : mid);
        size_t l1, l2;
        partition(ind, count, axis, cut, l1, l2);
        const size_t half = count / 2;
        const size_t k = l1 > half ? l1 : (l2 < half ? l2 : half);
        Span lb[3] = {box[0], box[1], box[2]}, rb[3] = {box[0], box[1], box[2]};
        lb[axis].hi = cut;
        const int c_lo = split(first, first + k, lb);
        rb[axis].lo = cut;
        const int c_hi = split(first + k, last, rb);
        Node& nd = nodes_[(size_t)me];
        nd.near_lo = c_lo; nd.near_hi = c_hi; nd.axis = axis; nd.lo_edge = lb[axis].hi; nd.hi_edge = rb[axis].lo;
        for (int a = 0; a < 3; a++) { box[a].lo = lb[a].lo < rb[a].lo ? lb[a].lo : rb[a].lo; box[a].hi = lb[a].hi > rb[a].hi ? lb[a].hi : rb[a].hi; }
        return me;
    }

    // searchLevel, nanoflann.hpp:1137-1184 (eps = 0)
    void descend(int node, const float q[3], float boxdist, float side[3], Best& b) const {
        const Node& nd = nodes_[(size_t)node];
        if (nd.near_lo < 0) {
            const float gate = b.d;                                 // sampled once per leaf (nanoflann.hpp:1144)
            for (size_t i = nd.first; i < nd.last; i++) {
                const int m = order_[i];
                const float e0 = q[0] - at(m, 0), e1 = q[1] - at(m, 1), e2 = q[2] - at(m, 2);
                const float d = e0 * e0 + e1 * e1 + e2 * e2;        // jly_icp3d.hpp:54-60
                if (d < gate && b.d > d) { b.d = d; b.id = m; }
            }
            return;
        }
        const float v = q[nd.axis];
        const float below = v - nd.lo_edge, above = v - nd.hi_edge;
        const bool low_first = (below + above) < 0;
        const float gap = low_first ? (v - nd.hi_edge) * (v - nd.hi_edge) : (v - nd.lo_edge) * (v - nd.lo_edge);
        descend(low_first ? nd.near_lo : nd.near_hi, q, boxdist, side, b);
        const float keep = side[nd.axis];
        boxdist = boxdist + gap - keep;
        side[nd.axis] = gap;
        if (boxdist * 1.0f <= b.d) descend(low_first ? nd.near_hi : nd.near_lo, q, boxdist, side, b);
        side[nd.axis] = keep;
    }

    bool built_ = false;
    int root_ = -1;
    Span box_[3] = {{0, 0}, {0, 0}, {0, 0}};
    std::vector<float> xyz_;
    std::vector<int> order_;
    std::vector<Node> nodes_;
};

}  // namespace goicp
#endif
