"""Run configuration: the same global dict, same keys and defaults as the reference's caesar_yolo/config.py:4-59
(filled by scripts/run.py:311-338), plus the device-path knobs this build adds (marked NEW)."""

CONFIG = {
    # detection
    'img_size': 640,
    'preprocess_fcn': None,
    'image_path': '',
    'image_xmin': 0, 'image_xmax': 0, 'image_ymin': 0, 'image_ymax': 0,
    # tiling
    'mpi': None,                       # kept for interface compatibility; ranks come from torch.distributed here
    'split_image_in_tiles': False,
    'tile_xsize': 256, 'tile_ysize': 256,
    'tile_xstep': 1.0, 'tile_ystep': 1.0,
    'max_ntasks_per_worker': None,     # reference: 100 (its sequential engine refuses more tiles per rank, inference.py:1151-1160);
                                       # the batched engine has no limit: the guard applies only when a number is given
    # source finding
    'devices': ['cpu'],
    'use_multi_gpu': False,
    'iou_thr': 0.5,
    'merge_overlap_iou_thr_soft': 0.3,
    'merge_overlap_iou_thr_hard': 0.8,
    'score_thr': 0.7,
    # outputs
    'save_catalog': True, 'save_tile_catalog': False, 'outfile_json': '',
    'save_region': True, 'save_tile_region': False, 'outfile': '',
    'save_img': False, 'save_tile_img': False,
    'draw_plot': False, 'draw_class_label_in_caption': True, 'save_plot': False,
    # NEW: HIP tile pipeline
    'tile_batch': 64,                  # tiles per kernel launch sequence
    'precision': 'fp16x3',             # 'fp16x3' (default: parity context), 'fp32' (exact fp32, ~2.7x slower), 'fp16' (throughput, ~3x faster)
}
