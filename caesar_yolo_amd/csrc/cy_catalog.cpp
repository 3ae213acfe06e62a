// Host-side catalog records and cross-tile merge (no GPU work: O(10^3-10^5) boxes once per mosaic).
//   cy_make_tile_records : Analyzer.make_json_results (caesar_yolo/evaluation.py:418-469) +
//                          SFinder.find_sources_at_edge (caesar_yolo/inference.py:663-726)
//   cy_merge_edge_sources: SFinder.merge_edge_sources (caesar_yolo/inference.py:731-931) with
//                          utils.get_merged_bbox (caesar_yolo/utils.py:110-119) and Graph (caesar_yolo/graph.py:2-41)
// The reference walks all source pairs and all tile pairs in Python (and logs each one); here neighbour tiles come from
// the same inclusive-range test (inference.py:123-163) evaluated per tile pair once, and pair tests only run between
// sources of neighbouring tiles.  Component order, survivor choice and naming order are the reference's.
#include "../../include/caesar_yolo_hip.h"
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {
inline bool tiles_neighbors(const int* a, const int* b) {
    const int ax0 = a[0], ax1 = a[1], ay0 = a[2], ay1 = a[3], bx0 = b[0], bx1 = b[1], by0 = b[2], by1 = b[3];
    const bool adjx = (ax1 == bx0 - 1) || (ax0 == bx1 + 1) || (ax0 == bx0 && ax1 == bx1);
    const bool adjy = (ay1 == by0 - 1) || (ay0 == by1 + 1) || (ay0 == by0 && ay1 == by1);
    const bool ovl = !(ax1 < bx0 || ax0 > bx1 || ay1 < by0 || ay0 > by1);
    return (adjx && adjy) || ovl;
}
// neighbour tile ids per tile (ascending), only for tiles flagged in `need`.
// The pair test factorises per axis -- neighbour = (adjX && adjY) || (ovlX && ovlY) -- so it is evaluated between the
// DISTINCT x-ranges and y-ranges of the grid (tens, not thousands) and expanded through a (x-range, y-range) -> tiles
// table: O(T * neighbours) instead of the reference's O(T^2) pair loop (inference.py:1034-1071), same lists.
std::vector<std::vector<int>> neighbor_lists(const int* tiles, int T, const std::vector<char>& need) {
    std::vector<std::vector<int>> nb(T);
    std::vector<std::pair<int, int>> xs, ys;
    std::vector<int> xi(T), yi(T);
    auto index_of = [](std::vector<std::pair<int, int>>& v, std::pair<int, int> r) {
        for (size_t k = 0; k < v.size(); ++k) if (v[k] == r) return (int)k;
        v.push_back(r);
        return (int)v.size() - 1;
    };
    for (int i = 0; i < T; ++i) {
        xi[i] = index_of(xs, {tiles[4 * i], tiles[4 * i + 1]});
        yi[i] = index_of(ys, {tiles[4 * i + 2], tiles[4 * i + 3]});
    }
    const int NX = (int)xs.size(), NY = (int)ys.size();
    if ((long)NX * NY > 4L * T + 64) {                       // irregular tile set: plain pair loop
        for (int i = 0; i < T; ++i) {
            if (!need[i]) continue;
            for (int j = 0; j < T; ++j)
                if (j != i && tiles_neighbors(tiles + 4 * i, tiles + 4 * j)) nb[i].push_back(j);
        }
        return nb;
    }
    auto adj = [](std::pair<int, int> a, std::pair<int, int> b) { return a.second == b.first - 1 || a.first == b.second + 1 || a == b; };
    auto ovl = [](std::pair<int, int> a, std::pair<int, int> b) { return !(a.second < b.first || a.first > b.second); };
    std::vector<std::vector<int>> cell((size_t)NX * NY);
    for (int i = 0; i < T; ++i) cell[(size_t)xi[i] * NY + yi[i]].push_back(i);
    std::vector<std::vector<std::pair<int, int>>> relx(NX), rely(NY);   // (other index, bit0 = adjacent | bit1 = overlapping)
    for (int a = 0; a < NX; ++a) for (int b = 0; b < NX; ++b) { const int f = (adj(xs[a], xs[b]) ? 1 : 0) | (ovl(xs[a], xs[b]) ? 2 : 0); if (f) relx[a].push_back({b, f}); }
    for (int a = 0; a < NY; ++a) for (int b = 0; b < NY; ++b) { const int f = (adj(ys[a], ys[b]) ? 1 : 0) | (ovl(ys[a], ys[b]) ? 2 : 0); if (f) rely[a].push_back({b, f}); }
    for (int i = 0; i < T; ++i) {
        if (!need[i]) continue;
        for (auto& rx : relx[xi[i]])
            for (auto& ry : rely[yi[i]]) {
                if (!((rx.second & ry.second & 1) || (rx.second & ry.second & 2))) continue;
                for (int j : cell[(size_t)rx.first * NY + ry.first]) if (j != i) nb[i].push_back(j);
            }
        std::sort(nb[i].begin(), nb[i].end());
    }
    return nb;
}
// Both entry points run on the same grid, once per mosaic pass: the lists are computed for every tile once and kept while
// the tile array is unchanged (compared by value, 16 bytes per tile).
const std::vector<std::vector<int>>& cached_neighbor_lists(const int* tiles, int T) {
    static thread_local std::vector<int> key;
    static thread_local std::vector<std::vector<int>> lists;
    if ((int)key.size() != 4 * T || !std::equal(key.begin(), key.end(), tiles)) {
        lists = neighbor_lists(tiles, T, std::vector<char>(T, 1));
        key.assign(tiles, tiles + 4 * (size_t)T);
    }
    return lists;
}
// Run f(lo, hi) over [0, n) on a few host threads (the two loops below are independent per element; at N GPUs this merge is
// the serial tail of the step on rank 0, so its 1-2 ms matter).  Small inputs stay on the calling thread.
constexpr int kMaxMergeThreads = 32;
template <typename F>
void parallel_ranges(int n, int min_per_thread, F f) {
    static const int forced = getenv("CY_MERGE_THREADS") ? atoi(getenv("CY_MERGE_THREADS")) : 0;   // an explicit count is taken as is
    int nt = forced > 0 ? forced : (int)std::thread::hardware_concurrency();
    if (forced <= 0 && nt > 4) nt = 4;
    if (nt > kMaxMergeThreads) nt = kMaxMergeThreads;        // callers size their per-thread slots with this
    if (nt < 1) nt = 1;
    if (n / (min_per_thread > 0 ? min_per_thread : 1) < nt) nt = n / (min_per_thread > 0 ? min_per_thread : 1);
    if (nt <= 1) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int lo = (int)((long)n * t / nt), hi = (int)((long)n * (t + 1) / nt);
        th.emplace_back([=]() { f(lo, hi, t); });
    }
    for (auto& x : th) x.join();
}
}  // namespace

extern "C" int cy_make_tile_records(const float* det, const int* det_tile, int n, const int* tiles, int T, double* rec) {
    if (n < 0 || T < 1 || (n > 0 && (!det || !det_tile || !rec)) || !tiles) return CY_ERR_ARG;
    for (int i = 0; i < n; ++i) if (det_tile[i] < 0 || det_tile[i] >= T) return CY_ERR_ARG;
    const auto& nb = cached_neighbor_lists(tiles, T);
    parallel_ranges(n, 2048, [&](int lo, int hi, int) {
    for (int i = lo; i < hi; ++i) {
        const int t = det_tile[i];
        const int* tc = tiles + 4 * t;
        const int nx = tc[1] - tc[0], ny = tc[3] - tc[2];
        const int x1 = (int)det[6 * i], y1 = (int)det[6 * i + 1], x2 = (int)det[6 * i + 2], y2 = (int)det[6 * i + 3];
        int edge = 0;
        if (x1 <= 0 || x1 >= nx - 1 || x2 <= 0 || x2 >= nx - 1) edge = 1;
        if (y1 <= 0 || y1 >= ny - 1 || y2 <= 0 || y2 >= ny - 1) edge = 1;
        const double gx1 = tc[0] + x1, gx2 = tc[0] + x2, gy1 = tc[2] + y1, gy2 = tc[2] + y2;
        if ((gx1 == tc[0] || gx2 == tc[1]) || (gy1 == tc[2] || gy2 == tc[3])) edge = 2;
        else {
            for (int j : nb[t]) {
                const int* q = tiles + 4 * j;
                if (gx2 < q[0] || gx1 > q[1] || gy2 < q[2] || gy1 > q[3]) continue;
                edge = 2;
                break;
            }
        }
        double* r = rec + 8 * i;
        r[0] = gx1; r[1] = gy1; r[2] = gx2; r[3] = gy2; r[4] = (double)det[6 * i + 4]; r[5] = (double)(int)det[6 * i + 5];
        r[6] = t; r[7] = edge;
    }
    });
    return CY_OK;
}

extern "C" int cy_merge_edge_sources(const double* rec, int n, const int* tiles, int T, double* out) {
    if (n < 0 || T < 1 || (n > 0 && (!rec || !out)) || !tiles) return CY_ERR_ARG;
    int nout = 0;
    std::vector<int> tbm;                       // indices of edge sources, in tile/source order
    for (int i = 0; i < n; ++i) {
        const double* r = rec + 8 * i;
        if (r[7] == 0.0) {
            double* o = out + 8 * nout++;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[4]; o[5] = r[5]; o[6] = 0.0; o[7] = 0.0;
        } else tbm.push_back(i);
    }
    const int N = (int)tbm.size();
    if (N == 0) return nout;
    // sources of each tile (positions in tbm are ascending because records are grouped by ascending tile): CSR
    std::vector<int> tstart(T + 1, 0);
    for (int k = 0; k < N; ++k) { const int t = (int)rec[8 * tbm[k] + 6]; if (t < 0 || t >= T) return CY_ERR_ARG; tstart[t + 1]++; }
    for (int t = 0; t < T; ++t) tstart[t + 1] += tstart[t];
    for (int k = 1; k < N; ++k) if (rec[8 * tbm[k] + 6] < rec[8 * tbm[k - 1] + 6]) return CY_ERR_ARG;   // must be tile-ordered
    const auto& nb = cached_neighbor_lists(tiles, T);
    // overlapping pairs (i < j) between sources of neighbouring tiles, generated in lexicographic order, so the CSR rows
    // below come out ascending == the reference's adjacency insertion order (i asc, then j asc)
    std::vector<std::pair<int, int>> pairs;
    std::vector<int> deg(N + 1, 0);
    std::vector<double> box(4 * (size_t)N);       // compact copy of the edge sources' boxes (cache-friendly pair loop)
    std::vector<int> tof(N);
    for (int k = 0; k < N; ++k) { const double* r = rec + 8 * tbm[k]; box[4 * k] = r[0]; box[4 * k + 1] = r[1]; box[4 * k + 2] = r[2]; box[4 * k + 3] = r[3]; tof[k] = (int)r[6]; }
    // ranges of i on a few threads, each with its own pair list; concatenated in range order they are in the same
    // lexicographic order a single loop produces
    std::vector<std::vector<std::pair<int, int>>> part(kMaxMergeThreads);
    parallel_ranges(N, 1024, [&](int lo, int hi, int slot) {
        auto& out_pairs = part[slot];
        for (int i = lo; i < hi; ++i) {
            const double ax1 = box[4 * i], ay1 = box[4 * i + 1], ax2 = box[4 * i + 2], ay2 = box[4 * i + 3];
            const int ti = tof[i];
            for (int tj : nb[ti]) {               // tid_j in neighborTileIds(tile_i): a tile is never its own neighbour
                if (tj < ti) continue;            // j > i implies tile_j >= tile_i
                // a source lies inside its tile's inclusive bounds (cy_make_tile_records), so a box that misses tile tj
                // misses every source of tile tj: skips most of the eight neighbours without touching their sources
                const int* q = tiles + 4 * tj;
                if (ax2 < q[0] || ax1 > q[1] || ay2 < q[2] || ay1 > q[3]) continue;
                for (int j = tstart[tj]; j < tstart[tj + 1]; ++j) {
                    if (j <= i) continue;
                    const double* b = &box[4 * j];
                    if (ax2 < b[0] || ax1 > b[2] || ay2 < b[1] || ay1 > b[3]) continue;
                    out_pairs.push_back({i, j});
                }
            }
        }
    });
    for (auto& v : part) for (auto& pr : v) { pairs.push_back(pr); deg[pr.first + 1]++; deg[pr.second + 1]++; }
    for (int v = 0; v < N; ++v) deg[v + 1] += deg[v];
    std::vector<int> adjv(pairs.size() * 2), fill(deg.begin(), deg.end() - 1);
    for (auto& pr : pairs) { adjv[fill[pr.first]++] = pr.second; adjv[fill[pr.second]++] = pr.first; }
    std::vector<char> vis(N, 0);
    std::vector<int> comp, stk, cur(deg.begin(), deg.end() - 1);
    for (int v0 = 0; v0 < N; ++v0) {
        if (vis[v0]) continue;
        comp.clear(); stk.clear();
        vis[v0] = 1; comp.push_back(v0); stk.push_back(v0);
        while (!stk.empty()) {                    // iterative form of the recursive DFS preorder
            const int v = stk.back();
            bool pushed = false;
            while (cur[v] < deg[v + 1]) {
                const int u = adjv[cur[v]++];
                if (!vis[u]) { vis[u] = 1; comp.push_back(u); stk.push_back(u); pushed = true; break; }
            }
            if (!pushed) stk.pop_back();
        }
        double* o = out + 8 * nout++;
        if (comp.size() == 1) {
            const double* r = rec + 8 * tbm[comp[0]];
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[4]; o[5] = r[5]; o[6] = r[7]; o[7] = 0.0;
            continue;
        }
        int ilarge = -1; double alarge = -1.0;
        double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        for (size_t k = 0; k < comp.size(); ++k) {
            const double* r = rec + 8 * tbm[comp[k]];
            const double area = (r[2] - r[0]) * (r[3] - r[1]);
            if (area > alarge) { alarge = area; ilarge = comp[k]; }
            if (k == 0) { x1 = r[0]; y1 = r[1]; x2 = r[2]; y2 = r[3]; }
            else { x1 = std::min(x1, r[0]); y1 = std::min(y1, r[1]); x2 = std::max(x2, r[2]); y2 = std::max(y2, r[3]); }
        }
        const double* big = rec + 8 * tbm[ilarge];
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = big[4]; o[5] = big[5]; o[6] = 2.0; o[7] = 1.0;
    }
    return nout;
}
