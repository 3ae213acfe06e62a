"""Turn the FP16DELTA lines of a `pytest tests/test_gpu_configs.py -s` log into the markdown table of DESIGN.md section 2
(developer tool).  python tools/fp16_delta.py gpurun_out/<log>"""
import json
import sys

rows = [json.loads(ln.split("FP16DELTA ", 1)[1]) for ln in open(sys.argv[1]) if "FP16DELTA " in ln]
print("| config | kept anchors after NMS: oracle / fp16 / common | box delta, same anchor (px; strides) | score delta, same anchor | "
      "per-tile detections after IoU merge: oracle / matched / missing / extra | catalog sources: oracle / matched / missing / extra | "
      "candidates within 1e-3 of the conf cut |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    n, t, c = r["nms"], r["per_tile"], r["catalog"]
    print("| %s | %d / %d / %d (%.1f %%) | %.2f; %.2f | %.1e | %d / %d / %d / %d (%.1f %%) | %d / %d / %d / %d (%.1f %%) | %d |" % (
        r["config"], n["ref"], n["got"], n["common"], 100.0 * n["common"] / max(n["ref"], 1), n["max_dbox"], n.get("max_dbox_strides", 0.0),
        n["max_dscore"], r["oracle_per_tile_detections"], t["matched"], t["missing"], t["extra"],
        100.0 * t["matched"] / max(r["oracle_per_tile_detections"], 1), r["oracle_sources"], c["matched"], c["missing"], c["extra"],
        100.0 * c["matched"] / max(r["oracle_sources"], 1), r["candidates_within_1e-3_of_conf"]))
