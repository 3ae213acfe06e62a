#!/usr/bin/env python
"""Command line of the detect path: same flags as the reference's scripts/run.py:58-155 (unused cosmetic flags are
accepted and ignored), same stage order for --preprocessing (:272-302), same CONFIG keys (:311-338).

Differences that come with the MI355X engine:
  --weights  takes a CYW1 file (caesar_yolo_amd/weights.py) or "seeded:<scale>:<nc>[:seed]";
  --devices  lists GPU indices ("0", "cuda:0", "0,1,2,3"); "cpu" (the reference's default) selects GPU LOCAL_RANK with a
             warning: there is no CPU path;
  --max_ntasks_per_worker  has no default here (the reference's 100 would refuse BASELINE configs 3-5); given, it is honoured;
  multi-GPU  = one process per GPU: `python -m torch.distributed.run --nproc-per-node N scripts/run.py ...`
             (replaces `mpirun -np N`, test/run_inference_parallel.sh:47-52);
  --precision fp16x3|fp32|fp16 (default fp16x3 = the fast parity context: fp16 high + low halves, fp32 accumulate; fp32 = exact fp32
  FMA chains, 2.7x slower; fp16 = throughput mode, 3x faster, ~2 % of detections differ), --tile_batch N  are new.
"""
import argparse
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caesar_yolo_amd.config import CONFIG                                     # noqa: E402
from caesar_yolo_amd.preprocessing import (DataPreprocessor, BkgSubtractor, SigmaClipShifter, SigmaClipper, ChanResizer,  # noqa: E402
                                           ZScaleTransformer, Chan3Trasformer, MinMaxNormalizer)
from caesar_yolo_amd.inference import SFinder                                 # noqa: E402
from caesar_yolo_amd.model import YOLO                                        # noqa: E402

logging.basicConfig(format="%(asctime)-15s %(levelname)s - %(message)s", datefmt='%Y-%m-%d %H:%M:%S')
logger = logging.getLogger("caesar_yolo_amd")
logger.setLevel(logging.INFO)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='CAESAR-YOLO options (MI355X build)')
    p.add_argument('--image', required=False, type=str)
    p.add_argument('--datalist', required=False)
    p.add_argument('--maxnimgs', required=False, type=int, default=-1)
    p.add_argument('--weights', required=True)
    p.add_argument('--imgsize', dest='imgsize', type=int, default=640)
    p.add_argument('--preprocessing', dest='preprocessing', action='store_true')
    p.add_argument('--normalize_minmax', dest='normalize_minmax', action='store_true')
    p.add_argument('-norm_min', '--norm_min', dest='norm_min', type=float, default=0.)
    p.add_argument('-norm_max', '--norm_max', dest='norm_max', type=float, default=1.)
    p.add_argument('--subtract_bkg', dest='subtract_bkg', action='store_true')
    p.add_argument('-sigma_bkg', '--sigma_bkg', dest='sigma_bkg', type=float, default=3)
    p.add_argument('--use_box_mask_in_bkg', dest='use_box_mask_in_bkg', action='store_true')
    p.add_argument('-bkg_box_mask_fract', '--bkg_box_mask_fract', dest='bkg_box_mask_fract', type=float, default=0.7)
    p.add_argument('-bkg_chid', '--bkg_chid', dest='bkg_chid', type=int, default=-1)
    p.add_argument('--clip_shift_data', dest='clip_shift_data', action='store_true')
    p.add_argument('-sigma_clip', '--sigma_clip', dest='sigma_clip', type=float, default=1)
    p.add_argument('--clip_data', dest='clip_data', action='store_true')
    p.add_argument('-sigma_clip_low', '--sigma_clip_low', dest='sigma_clip_low', type=float, default=10)
    p.add_argument('-sigma_clip_up', '--sigma_clip_up', dest='sigma_clip_up', type=float, default=10)
    p.add_argument('-clip_chid', '--clip_chid', dest='clip_chid', type=int, default=-1)
    p.add_argument('--zscale_stretch', dest='zscale_stretch', action='store_true')
    p.add_argument('--zscale_contrasts', dest='zscale_contrasts', type=str, default='0.25,0.25,0.25')
    p.add_argument('--chan3_preproc', dest='chan3_preproc', action='store_true')
    p.add_argument('-sigma_clip_baseline', '--sigma_clip_baseline', dest='sigma_clip_baseline', type=float, default=0)
    p.add_argument('-nchannels', '--nchannels', dest='nchannels', type=int, default=1)
    p.add_argument('--scoreThr', default=0.7, type=float)
    p.add_argument('--iouThr', default=0.5, type=float)
    p.add_argument('--merge_overlap_iou_thr_soft', default=0.3, type=float)
    p.add_argument('--merge_overlap_iou_thr_hard', default=0.8, type=float)
    for a in ('xmin', 'xmax', 'ymin', 'ymax'):
        p.add_argument('--' + a, dest=a, type=int, default=-1)
    p.add_argument('--split_img_in_tiles', dest='split_img_in_tiles', action='store_true')
    p.add_argument('--tile_xsize', type=int, default=512)
    p.add_argument('--tile_ysize', type=int, default=512)
    p.add_argument('--tile_xstep', type=float, default=1.0)
    p.add_argument('--tile_ystep', type=float, default=1.0)
    p.add_argument('--max_ntasks_per_worker', type=int, default=None,
                   help='refuse the run if a rank holds more tiles (reference default 100; here: no limit unless given)')
    p.add_argument('--devices', type=str, default="cpu")
    p.add_argument('--multigpu', dest='multigpu', action='store_true')
    for a in ('draw_plots', 'draw_class_label_in_caption', 'save_plots', 'save_tile_catalog', 'save_tile_region', 'save_tile_img'):
        p.add_argument('--' + a, dest=a, action='store_true')
    p.add_argument('--detect_outfile', type=str, default="")
    p.add_argument('--detect_outfile_json', type=str, default="")
    p.add_argument('--precision', type=str, default="fp16x3", choices=["fp16", "fp16x3", "fp32"],
                   help='arithmetic of the detector.  fp16x3 (default): parity context -- activations and weights as fp16 high + low halves, '
                        'fp32 accumulate; kept boxes identical to the fp32 oracle, boxes within 1e-4 of the image size (<= 0.06 px), scores 2e-5.  '
                        'fp32: exact fp32 FMA chains (the reference\'s own arithmetic), ~2.7x slower than fp16x3.  fp16: throughput mode, fp16 '
                        'operands / fp32 accumulate, ~3x the fp16x3 rate; 2-4 %% of the detections differ from the fp32 run (DESIGN.md section 2)')
    p.add_argument('--tile_batch', type=int, default=64)
    return p.parse_args(argv)


def validate_args(args):
    if not args.image:
        logger.error("Argument --image is required for detect task!")
        return -1
    if not os.path.isfile(args.image):
        logger.error("Image argument must be an existing image on filesystem!")
        return -1
    if not args.image.endswith('.fits'):
        logger.error("Image must have .fits extension on the HIP path!")
        return -1
    if not args.weights.startswith("seeded:") and not os.path.isfile(args.weights):
        logger.error("Given weight file %s not existing or not a file!" % args.weights)
        return -1
    if args.split_img_in_tiles and (args.xmin >= 0 or args.xmax >= 0 or args.ymin >= 0 or args.ymax >= 0):
        # serial runs crop like the reference (inference.py:499-505); the tiled run of the reference derives its grid from
        # attributes it has not set yet when a range is given (inference.py:374-381): refused here
        logger.error("Sub-image ranges together with --split_img_in_tiles are not supported (broken in the reference: inference.py:374-381)")
        return -1
    return 0


def build_preprocessor(args):
    """Stage list in the reference's fixed order (scripts/run.py:272-302)."""
    zc = [float(x) for x in args.zscale_contrasts.split(',')]
    st = []
    if args.subtract_bkg:
        st.append(BkgSubtractor(sigma=args.sigma_bkg, use_mask_box=args.use_box_mask_in_bkg, mask_fract=args.bkg_box_mask_fract, chid=args.bkg_chid))
    if args.clip_shift_data:
        st.append(SigmaClipShifter(sigma=args.sigma_clip, chid=args.clip_chid))
    if args.clip_data:
        st.append(SigmaClipper(sigma_low=args.sigma_clip_low, sigma_up=args.sigma_clip_up, chid=args.clip_chid))
    if args.nchannels > 1:
        st.append(ChanResizer(nchans=args.nchannels))
    if args.zscale_stretch:
        st.append(ZScaleTransformer(contrasts=zc))
    if args.chan3_preproc:
        st.append(Chan3Trasformer(sigma_clip_baseline=args.sigma_clip_baseline, sigma_clip_low=args.sigma_clip_low,
                                  sigma_clip_up=args.sigma_clip_up, zscale_contrast=zc[0]))
    if args.normalize_minmax:
        st.append(MinMaxNormalizer(norm_min=args.norm_min, norm_max=args.norm_max))
    if not args.preprocessing:
        return None
    if not st:
        logger.warning("No pre-processing steps defined ...")
        return None
    return DataPreprocessor(st)


def main(argv=None):
    args = parse_args(argv)
    if validate_args(args) < 0:
        return 1
    if args.chan3_preproc and args.nchannels != 3:
        logger.error("You selected chan3_preproc pre-processing options, you must set nchannels options to 3!")
        return 1
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    C = CONFIG
    C.update({'img_size': args.imgsize, 'preprocess_fcn': build_preprocessor(args), 'image_path': args.image,
              'image_xmin': args.xmin, 'image_xmax': args.xmax, 'image_ymin': args.ymin, 'image_ymax': args.ymax,
              'split_image_in_tiles': args.split_img_in_tiles, 'tile_xsize': args.tile_xsize, 'tile_ysize': args.tile_ysize,
              'tile_xstep': args.tile_xstep, 'tile_ystep': args.tile_ystep, 'max_ntasks_per_worker': args.max_ntasks_per_worker,
              'devices': [str(x) for x in args.devices.split(',')], 'use_multi_gpu': args.multigpu, 'iou_thr': args.iouThr,
              'score_thr': args.scoreThr, 'merge_overlap_iou_thr_soft': args.merge_overlap_iou_thr_soft,
              'merge_overlap_iou_thr_hard': args.merge_overlap_iou_thr_hard, 'outfile': args.detect_outfile,
              'outfile_json': args.detect_outfile_json, 'save_region': True, 'tile_batch': args.tile_batch,
              'draw_plot': args.draw_plots, 'draw_class_label_in_caption': args.draw_class_label_in_caption,
              'save_plot': args.save_plots,
              'save_tile_catalog': args.save_tile_catalog, 'save_tile_region': args.save_tile_region,
              'save_tile_img': args.save_tile_img,
              'precision': args.precision})
    model = YOLO(args.weights, precision=args.precision, max_batch=args.tile_batch if args.split_img_in_tiles else 1,
                 max_imgsz=max(args.imgsize, 32))
    sfinder = SFinder(model, C)
    status = sfinder.run_parallel() if args.split_img_in_tiles else sfinder.run()
    if world > 1:
        dist.destroy_process_group()
    if status < 0:
        logger.error("sfinder run failed, see logs...")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
