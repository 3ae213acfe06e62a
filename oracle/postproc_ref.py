"""CPU oracle for tiling, per-tile IoU merge, edge flagging, cross-tile merge and the catalog.
TEST INFRASTRUCTURE ONLY (importers: tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

Restates (paths relative to /root/reference):
  caesar_yolo/utils.py:622-697        generate_tiles
  caesar_yolo/utils.py:54-107         get_iou
  caesar_yolo/utils.py:110-119        get_merged_bbox
  caesar_yolo/graph.py:2-41           Graph (adjacency lists + recursive DFS preorder)
  caesar_yolo/evaluation.py:252-346   Analyzer.process_detections
  caesar_yolo/evaluation.py:418-469   Analyzer.make_json_results
  caesar_yolo/inference.py:123-163    TileTask.is_task_tile_{adjacent,overlapping,neighbor}
  caesar_yolo/inference.py:992-1071   SFinder.create_tile_tasks (assignment + neighbour lists)
  caesar_yolo/inference.py:663-726    SFinder.find_sources_at_edge
  caesar_yolo/inference.py:731-931    SFinder.merge_edge_sources
  caesar_yolo/inference.py:1197-1211  write_json_results (json.dump indent=2 sort_keys=True)

Pinned by tests/golden/{tiles,neighbors}.json, process_detections.npz, catalog_*.json, all
captured from the imported reference by oracle/gen_golden.py (numpy 1.26: np.float32 scalar
arithmetic promotes to float64 when mixed with a Python float -- that is why get_iou divides
in double here).
"""
import json
import numpy as np


# --------------------------------------------------------------------------- tiles
def generate_tiles(img_xmin, img_xmax, img_ymin, img_ymax, tsx, tsy, stepx, stepy):
    """utils.py:622-697.  Returns [(xmin, xmax_excl, ymin, ymax_excl)] y-outer/x-inner, or None."""
    if img_xmax <= img_xmin or img_ymax <= img_ymin:
        return None
    if tsx <= 0 or tsy <= 0:
        return None
    if stepx <= 0 or stepy <= 0 or stepx > 1 or stepy > 1:
        return None
    nx = img_xmax - img_xmin + 1
    ny = img_ymax - img_ymin + 1
    if tsx > nx or tsy > ny:
        return None
    sx = int(np.round(stepx * tsx))
    sy = int(np.round(stepy * tsy))

    def axis(n, ts, st):
        lo, hi, idx = [], [], 0
        while idx <= n:
            off = min(ts, n - idx)
            if idx >= n or off == 0:
                break
            lo.append(idx)
            hi.append(idx + off)
            idx += st
        return lo, hi
    ylo, yhi = axis(ny, tsy, sy)
    xlo, xhi = axis(nx, tsx, sx)
    return [(img_xmin + xlo[i], img_xmin + xhi[i], img_ymin + ylo[j], img_ymin + yhi[j])
            for j in range(len(ylo)) for i in range(len(xlo))]


def tiles_are_neighbors(a, b):
    """inference.py:123-163 with coords = (ix_min, ix_max, iy_min, iy_max), max exclusive but
    compared inclusively (so tiles that merely touch count as overlapping)."""
    ax0, ax1, ay0, ay1 = a
    bx0, bx1, by0, by1 = b
    adj_x = (ax1 == bx0 - 1) or (ax0 == bx1 + 1) or (ax0 == bx0 and ax1 == bx1)
    adj_y = (ay1 == by0 - 1) or (ay0 == by1 + 1) or (ay0 == by0 and ay1 == by1)
    overlapping = not (ax1 < bx0 or ax0 > bx1 or ay1 < by0 or ay0 > by1)
    return (adj_x and adj_y) or overlapping


def create_tile_tasks(grid, nproc=1):
    """inference.py:1008-1071: round-robin tid -> worker, then neighbour lists in the reference's
    discovery order (same-worker pairs first, then across workers).  Returns one dict per tid."""
    per_worker = [[] for _ in range(nproc)]
    tasks = []
    w = 0
    for tid, c in enumerate(grid):
        t = {"tid": tid, "wid": w, "windex": len(per_worker[w]), "coords": list(c),
             "neighborTaskId": [], "neighborTaskIndex": [], "neighborWorkerId": []}
        per_worker[w].append(t)
        tasks.append(t)
        w = 0 if w >= nproc - 1 else w + 1

    def link(a, b):
        a["neighborTaskId"].append(b["tid"]); a["neighborTaskIndex"].append(b["windex"]); a["neighborWorkerId"].append(b["wid"])
    for i in range(nproc):
        for j, task in enumerate(per_worker[i]):
            for k in range(j + 1, len(per_worker[i])):
                o = per_worker[i][k]
                if tiles_are_neighbors(task["coords"], o["coords"]):
                    link(task, o); link(o, task)
            for s in range(i + 1, nproc):
                for o in per_worker[s]:
                    if tiles_are_neighbors(task["coords"], o["coords"]):
                        link(task, o); link(o, task)
    return tasks


# --------------------------------------------------------------------------- graph
def connected_components(n, edges):
    """graph.py:2-41: adjacency in insertion order, DFS preorder from vertex 0 upward."""
    adj = [[] for _ in range(n)]
    for v, w in edges:
        adj[v].append(w)
        adj[w].append(v)
    visited = [False] * n
    cc = []
    for v0 in range(n):
        if visited[v0]:
            continue
        comp, stack = [], [(v0, 0)]
        visited[v0] = True
        comp.append(v0)
        while stack:                      # iterative form of the recursive preorder
            v, k = stack.pop()
            while k < len(adj[v]):
                u = adj[v][k]
                k += 1
                if not visited[u]:
                    visited[u] = True
                    comp.append(u)
                    stack.append((v, k))
                    stack.append((u, 0))
                    break
        cc.append(comp)
    return cc


# --------------------------------------------------------------------------- per-tile IoU merge
def get_iou(bb1, bb2):
    """utils.py:54-107 on np.float32 boxes: areas in float32, the division in float64."""
    f = np.float32
    x1a, y1a, x2a, y2a = (f(v) for v in bb1)
    x1b, y1b, x2b, y2b = (f(v) for v in bb2)
    assert x1a < x2a and y1a < y2a and x1b < x2b and y1b < y2b
    xl, yt = max(x1a, x1b), max(y1a, y1b)
    xr, yb = min(x2a, x2b), min(y2a, y2b)
    if xr < xl or yb < yt:
        return 0.0
    inter = f(f(xr - xl) * f(yb - yt))
    a1 = f(f(x2a - x1a) * f(y2a - y1a))
    a2 = f(f(x2b - x1b) * f(y2b - y1b))
    union = f(f(a1 + a2) - inter)
    return float(inter) / float(union)


def process_detections(xyxy, conf, cls, score_thr, soft, hard):
    """evaluation.py:252-346.  Returns (kept boxes [M,4] f32, scores [M] f32, class ids [M] int,
    indices into the INPUT arrays)."""
    xyxy = np.asarray(xyxy, np.float32).reshape(-1, 4)
    conf = np.asarray(conf, np.float32).reshape(-1)
    cls = np.asarray(cls).reshape(-1)
    sel = [i for i in range(len(conf)) if not (conf[i] < score_thr)]          # :282
    n = len(sel)
    edges = []
    for a in range(n - 1):
        for b in range(a + 1, n):
            same = int(cls[sel[a]]) == int(cls[sel[b]])
            iou = get_iou(xyxy[sel[a]], xyxy[sel[b]])
            if iou >= hard or (same and iou >= soft):
                edges.append((a, b))
    keep = []
    for comp in connected_components(n, edges):
        best, ibest = 0, -1
        for idx in comp:
            if conf[sel[idx]] > best:                                          # first-wins on ties
                best, ibest = conf[sel[idx]], idx
        keep.append(sel[ibest])
    keep = np.array(keep, np.int64)
    return xyxy[keep].reshape(-1, 4), conf[keep], cls[keep].astype(np.int32), keep


# --------------------------------------------------------------------------- catalog objects
def make_objs(xyxy, conf, cls, names, nx, ny, xmin=0, ymin=0, tag=""):
    """evaluation.py:418-469: int() truncation, tile-local edge rule, add tile origin."""
    objs = []
    for i in range(len(conf)):
        x1, y1, x2, y2 = (int(v) for v in xyxy[i])
        at_edge = (x1 <= 0 or x1 >= nx - 1 or x2 <= 0 or x2 >= nx - 1 or
                   y1 <= 0 or y1 >= ny - 1 or y2 <= 0 or y2 >= ny - 1)
        cid = int(cls[i])
        objs.append({"name": "S%d" % (i + 1) + ("_" + tag if tag else ""),
                     "x1": float(xmin + x1), "x2": float(xmin + x2),
                     "y1": float(ymin + y1), "y2": float(ymin + y2),
                     "class_id": cid, "class_name": str(names[cid]),
                     "score": float(conf[i]), "edge": int(at_edge)})
    return objs


def flag_edge_sources(objs, coords, neighbor_coords):
    """inference.py:663-726: edge=True if on the tile's own bounds, else if the bbox overlaps
    (inclusive) any neighbour tile's range."""
    xmin, xmax, ymin, ymax = coords
    for o in objs:
        if (o["x1"] == xmin or o["x2"] == xmax) or (o["y1"] == ymin or o["y2"] == ymax):
            o["edge"] = True
            continue
        for (nx0, nx1, ny0, ny1) in neighbor_coords:
            if o["x2"] < nx0 or o["x1"] > nx1 or o["y2"] < ny0 or o["y1"] > ny1:
                continue
            o["edge"] = True
            break


def merge_edge_sources(tile_sources):
    """inference.py:731-931.  tile_sources: list (tile order) of dicts with "objs", "tileId",
    "neighborTileIds".  Returns the final source list (renamed S1..SN)."""
    final, tbm = [], []
    for ti, td in enumerate(tile_sources):
        for sj, s in enumerate(td["objs"]):
            if not s["edge"]:
                s = dict(s); s["merged"] = False
                final.append(s)
            else:
                tbm.append((sj, ti))
    n = len(tbm)
    edges = []
    for i in range(n):
        si, ti = tbm[i]
        a = tile_sources[ti]["objs"][si]
        nb = tile_sources[ti]["neighborTileIds"]
        for j in range(i + 1, n):
            sj, tj = tbm[j]
            b = tile_sources[tj]["objs"][sj]
            if tile_sources[tj]["tileId"] not in nb:
                continue
            if a["x2"] < b["x1"] or a["x1"] > b["x2"] or a["y2"] < b["y1"] or a["y1"] > b["y2"]:
                continue
            edges.append((i, j))
    for ci, comp in enumerate(connected_components(n, edges)):
        if len(comp) == 1:
            sj, tj = tbm[comp[0]]
            s = dict(tile_sources[tj]["objs"][sj]); s["merged"] = False
            final.append(s)
            continue
        ilarge, alarge, boxes = -1, -1, []
        for idx in comp:
            sj, tj = tbm[idx]
            s = tile_sources[tj]["objs"][sj]
            area = (s["x2"] - s["x1"]) * (s["y2"] - s["y1"])
            if area > alarge:
                alarge, ilarge = area, idx
            boxes.append((s["x1"], s["y1"], s["x2"], s["y2"]))
        sj, tj = tbm[ilarge]
        big = tile_sources[tj]["objs"][sj]
        arr = np.array(boxes)
        final.append({"x1": float(arr[:, 0].min()), "y1": float(arr[:, 1].min()),
                      "x2": float(arr[:, 2].max()), "y2": float(arr[:, 3].max()),
                      "edge": True, "merged": True, "score": big["score"],
                      "class_name": big["class_name"], "class_id": big["class_id"]})
    for i, s in enumerate(final):
        s["name"] = "S%d" % (i + 1)
    return final


def catalog_text(obj):
    """json.dump(..., indent=2, sort_keys=True) as in evaluation.py:481-482 / inference.py:1210-1211."""
    return json.dumps(obj, indent=2, sort_keys=True)


def run_tiled_reference(grid, dets_per_tile, skipped_tids, names, thr, image_id):
    """Whole-catalog oracle for a tiled run at P=1 (inference.py:578-658) given the detector output
    per tile: process_detections -> make_objs -> flag_edge_sources -> merge_edge_sources."""
    tasks = create_tile_tasks(grid, 1)
    tile_sources = []
    for t in tasks:
        tid = t["tid"]
        if tid in skipped_tids:
            continue
        x0, x1, y0, y1 = t["coords"]
        b, s, c = dets_per_tile[tid]
        kb, ks, kc, _ = process_detections(b, s, c, thr["score_thr"], thr["soft"], thr["hard"])
        if len(ks) == 0:
            continue
        objs = make_objs(kb, ks, kc, names, x1 - x0, y1 - y0, x0, y0, "t%d" % tid)
        flag_edge_sources(objs, t["coords"], [tasks[k]["coords"] for k in t["neighborTaskId"]])
        tile_sources.append({"image_id": image_id, "objs": objs, "workerId": 0, "tileId": tid,
                             "neighborTileIds": t["neighborTaskId"], "xmin": x0, "xmax": x1,
                             "ymin": y0, "ymax": y1})
    return tile_sources, {"sources": merge_edge_sources(tile_sources)}
