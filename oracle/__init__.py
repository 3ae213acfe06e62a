"""CPU oracle: a restatement of the reference's tiled-detect hot path used ONLY as the checker.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (caesar_yolo_amd) must never import from here.
"""
