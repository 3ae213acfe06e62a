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
#include <vector>

namespace {
inline bool tiles_neighbors(const int* a, const int* b) {
    const int ax0 = a[0], ax1 = a[1], ay0 = a[2], ay1 = a[3], bx0 = b[0], bx1 = b[1], by0 = b[2], by1 = b[3];
    const bool adjx = (ax1 == bx0 - 1) || (ax0 == bx1 + 1) || (ax0 == bx0 && ax1 == bx1);
    const bool adjy = (ay1 == by0 - 1) || (ay0 == by1 + 1) || (ay0 == by0 && ay1 == by1);
    const bool ovl = !(ax1 < bx0 || ax0 > bx1 || ay1 < by0 || ay0 > by1);
    return (adjx && adjy) || ovl;
}
// neighbour tile ids per tile (ascending), only for tiles flagged in `need`
std::vector<std::vector<int>> neighbor_lists(const int* tiles, int T, const std::vector<char>& need) {
    std::vector<std::vector<int>> nb(T);
    for (int i = 0; i < T; ++i) {
        if (!need[i]) continue;
        for (int j = 0; j < T; ++j)
            if (j != i && tiles_neighbors(tiles + 4 * i, tiles + 4 * j)) nb[i].push_back(j);
    }
    return nb;
}
}  // namespace

extern "C" int cy_make_tile_records(const float* det, const int* det_tile, int n, const int* tiles, int T, double* rec) {
    if (n < 0 || T < 1 || (n > 0 && (!det || !det_tile || !rec)) || !tiles) return CY_ERR_ARG;
    std::vector<char> need(T, 0);
    for (int i = 0; i < n; ++i) { if (det_tile[i] < 0 || det_tile[i] >= T) return CY_ERR_ARG; need[det_tile[i]] = 1; }
    const auto nb = neighbor_lists(tiles, T, need);
    for (int i = 0; i < n; ++i) {
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
    // sources of each tile (positions in tbm are ascending because records are grouped by ascending tile)
    std::vector<std::vector<int>> by_tile(T);
    std::vector<char> need(T, 0);
    for (int k = 0; k < N; ++k) { const int t = (int)rec[8 * tbm[k] + 6]; if (t < 0 || t >= T) return CY_ERR_ARG; by_tile[t].push_back(k); need[t] = 1; }
    const auto nb = neighbor_lists(tiles, T, need);
    std::vector<std::vector<int>> adj(N);
    for (int i = 0; i < N; ++i) {
        const double* a = rec + 8 * tbm[i];
        const int ti = (int)a[6];
        for (int tj : nb[ti]) {                   // tid_j in neighborTileIds(tile_i): a tile is never its own neighbour
            for (int j : by_tile[tj]) {
                if (j <= i) continue;
                const double* b = rec + 8 * tbm[j];
                if (a[2] < b[0] || a[0] > b[2] || a[3] < b[1] || a[1] > b[3]) continue;
                adj[i].push_back(j); adj[j].push_back(i);
            }
        }
    }
    for (auto& v : adj) std::sort(v.begin(), v.end());     // == the reference's insertion order (i asc, then j asc)
    std::vector<char> vis(N, 0);
    std::vector<int> comp, stk, cur(N, 0);
    for (int v0 = 0; v0 < N; ++v0) {
        if (vis[v0]) continue;
        comp.clear(); stk.clear();
        vis[v0] = 1; comp.push_back(v0); stk.push_back(v0);
        while (!stk.empty()) {                    // iterative form of the recursive DFS preorder
            const int v = stk.back();
            bool pushed = false;
            while (cur[v] < (int)adj[v].size()) {
                const int u = adj[v][cur[v]++];
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
