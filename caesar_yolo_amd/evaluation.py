"""Per-frame detect object: host mirror of the reference's `Analyzer` (caesar_yolo/evaluation.py:38-556) on the HIP path.

`Analyzer(model, config).predict(image, image_id, header, xmin, ymin)` keeps the reference's contract: 0 / -1 return,
results in `.bboxes_final / .scores_final / .class_ids_final / .labels_final / .results`; the work between the image and
the merged boxes (3-channel cube, preprocessing, rejection checks, model call, score filter + IoU graph merge:
evaluation.py:146-346) is one `cy_detect_tiles` call with the frame as a single tile.  The optional outputs are mirrored
too: out_<id>.json (:472-482), the DS9 region file (:487-548, Analyzer's own colour map), the plot (`draw_results`,
:351-411, matplotlib) and the preprocessed image as FITS (`write_fits`, :550-554); plot and FITS show the PREPROCESSED
image, which is fetched from the device (`cy_preproc_planes`) only when one of them is asked for."""
import json
import logging
import numpy as np
import torch

from . import lib as L
from .inference import objs_from_detections
from .preprocessing import no_preprocessing

logger = logging.getLogger("caesar_yolo_amd")


class Analyzer(object):
    def __init__(self, model, config):
        self.model, self.config = model, config
        self.class_names = model.names
        self.n_classes = len(model.names)
        self.imgsize = config['img_size']
        self.device = config['devices'][0]
        self.iou_thr, self.score_thr = config['iou_thr'], config['score_thr']
        self.merge_overlap_iou_thr_soft = config['merge_overlap_iou_thr_soft']
        self.merge_overlap_iou_thr_hard = config['merge_overlap_iou_thr_hard']
        self.write_to_json = config.get('save_catalog', True)
        self.outfile_json = ""
        self.write_to_ds9 = config.get('save_region', True)      # caesar_yolo/evaluation.py:97
        self.outfile_ds9 = ""
        self.draw = config.get('draw_plot', False)                # :93-95
        self.save_plots = config.get('save_plot', False)
        self.draw_class_label_in_caption = config.get('draw_class_label_in_caption', True)
        self.outfile = ""
        self.save_img = config.get('save_img', False)            # :98
        self.outfile_img = ""
        self.image = None                                         # (H,W,3) float64 preprocessed image, when a plot / FITS was asked for
        self.class_color_map = {'bkg': (0, 0, 0), 'spurious': (1, 0, 0), 'compact': (0, 0, 1), 'extended': (1, 1, 0),
                                'extended-multisland': (1, 0.647, 0), 'flagged': (0, 0, 0)}       # :100-107
        self.obj_name_tag = ""
        self.image_id = -1
        self.image_xmin = self.image_ymin = 0
        self.bboxes_final, self.scores_final, self.class_ids_final, self.labels_final = [], [], [], []
        self.results = {}

    def predict(self, image, image_id='', header=None, xmin=0, ymin=0):
        if image is None:
            logger.error("No input image given!")
            return -1
        image = np.asarray(image)
        if image.ndim == 3 and image.shape[2] == 3:
            return self._predict_cube(image, image_id, xmin, ymin)      # caller-supplied 3-channel image (evaluation.py:146-154: taken as is)
        if image.ndim != 2:
            logger.error("Expected a 2-D frame or an (H,W,3) image, got shape %s" % (image.shape,))
            return -1
        if image_id:
            self.image_id = image_id
        self.image_xmin, self.image_ymin = xmin, ymin
        dev = None if str(self.device) == "" else self.device          # 'cpu' -> this process's GPU (model._dev_index)
        try:
            det = self.model.engine(dev)
            dp = self.config.get('preprocess_fcn')
            cfg = dp.program() if dp is not None else no_preprocessing()
            ny, nx = image.shape
            mosaic = det.mosaic_to_device(image)
            d, cnt, status = det.detect_tiles(mosaic, [(0, 0)], ny, nx, self.imgsize, cfg, self.score_thr, self.iou_thr,
                                              self.merge_overlap_iou_thr_soft, self.merge_overlap_iou_thr_hard)
            torch.cuda.synchronize(det.tdev)
        except L.CyError as e:
            logger.warning("Model prediction failed (err=%s)..." % str(e))
            return -1
        return self._finish(det, d, cnt, int(status[0]), nx, ny, xmin, ymin,
                            lambda: det.preproc_planes(mosaic, [(0, 0)], ny, nx, cfg)[0][0])

    def _predict_cube(self, image, image_id, xmin, ymin):
        """An (H,W,3) array whose channels differ (the reference takes it as is, caesar_yolo/evaluation.py:146-154, and every CLI
        stage maps channel c of its input to channel c of its output): each plane goes through ITS channel's program on the device
        (cy_preproc_planes), the three float64 results are the cube handed to the model's own LetterBox (cy_letterbox_pack) ->
        forward -> decode/NMS -> IoU merge.  Pixel values are taken as fp32, like FITS pixels; the constant-row check
        (:171-176, rows 0..2 of the cube, Q1) runs on the host over the three rows it reads."""
        import copy
        if image_id:
            self.image_id = image_id
        self.image_xmin, self.image_ymin = xmin, ymin
        dev = None if str(self.device) == "" else self.device
        ny, nx = image.shape[:2]
        try:
            det = self.model.engine(dev)
            dp = self.config.get('preprocess_fcn')
            cfg = dp.program() if dp is not None else no_preprocessing()
            planes, st = [], 0
            for c in range(3):
                cfg_c = copy.copy(cfg)
                if cfg.nprog == 3:                                 # this channel's program alone, replicated
                    cfg_c = type(cfg)()
                    cfg_c.nprog = 1
                    cfg_c.prog[0] = cfg.prog[c]
                mosaic = det.mosaic_to_device(np.ascontiguousarray(image[:, :, c], dtype=np.float32))
                pl, status = det.preproc_planes(mosaic, [(0, 0)], ny, nx, cfg_c)
                torch.cuda.synchronize(det.tdev)
                if int(status[0]) == 1:
                    st = 1
                planes.append(pl[0, 0])
            cube = torch.stack(planes, 0)[None].contiguous()        # [1,3,H,W] float64 on device (a copy, no arithmetic)
            if st == 0:
                rows = cube[0, :, :3, :].cpu().numpy()              # rows 0..2 of the (H,W,3) image = [3 ch, 3 rows, W]
                if any(rows[:, i, :].min() == rows[:, i, :].max() for i in range(min(3, ny))):
                    st = 2
            d = cnt = None
            if st == 0:
                netin, lb = det.letterbox_pack(cube, self.imgsize)
                pred = det.forward(netin)
                dn, _, cn = det.decode_nms(pred, lb.H, lb.W, ny, nx, self.score_thr, self.iou_thr)
                d, cnt, _ = det.iou_merge(dn, cn, self.score_thr, self.merge_overlap_iou_thr_soft, self.merge_overlap_iou_thr_hard)
                torch.cuda.synchronize(det.tdev)
        except L.CyError as e:
            logger.warning("Model prediction failed (err=%s)..." % str(e))
            return -1
        return self._finish(det, d, cnt, st, nx, ny, xmin, ymin, lambda: cube[0])

    def _finish(self, det, d, cnt, st, nx, ny, xmin, ymin, planes_fn):
        if st == 1:
            logger.warning("Input image is None, no prediction made.")
            return -1
        if st == 2:
            logger.warning("Input image pixels have the same value in one of the first rows, no prediction made.")
            return -1
        if self.draw or self.save_img:
            self.image = np.ascontiguousarray(planes_fn().cpu().numpy().transpose(1, 2, 0))
        dd = d[0, :int(cnt[0])].cpu().numpy()
        self.bboxes_final = [dd[i, :4].copy() for i in range(dd.shape[0])]
        self.scores_final = [dd[i, 4] for i in range(dd.shape[0])]
        self.class_ids_final = [int(dd[i, 5]) for i in range(dd.shape[0])]
        self.labels_final = [self.class_names[c] for c in self.class_ids_final]
        self.results = {"image_id": self.image_id,
                        "objs": objs_from_detections(dd, self.class_names, nx, ny, xmin, ymin, self.obj_name_tag)}
        if self.draw:                                             # caesar_yolo/evaluation.py:203-210
            self.draw_results(self.outfile or ('out_' + str(self.image_id) + '.png'))
        if self.write_to_json:
            self.write_json_results(self.outfile_json or ('out_' + str(self.image_id) + '.json'))
        if self.write_to_ds9:                                     # caesar_yolo/evaluation.py:228-234
            self.write_ds9_regions(self.outfile_ds9 or ('out_' + str(self.image_id) + '.reg'))
        if self.save_img:                                         # :237-243
            self.write_fits(self.outfile_img or ('out_' + str(self.image_id) + '.fits'))
        return 0

    def write_ds9_regions(self, outfile):
        from . import utils
        utils.write_ds9_regions(outfile, self.results.get("objs", []), color_map=utils.CLASS_COLOR_MAP_DS9_FRAME, merged_tag=False)

    def write_fits(self, outfile):
        """caesar_yolo/evaluation.py:550-554: channel 0 of the preprocessed image, float64."""
        from . import utils
        logger.info("Saving 2D image to file %s ..." % outfile)
        utils.write_fits_image(outfile, self.image[:, :, 0], bitpix=-64)

    def draw_results(self, outfile):
        """caesar_yolo/evaluation.py:351-411: the preprocessed image with one rectangle + caption per final detection; written
        to `outfile` when save_plots is set, shown otherwise (needs matplotlib; a missing matplotlib is a warning, not an error)."""
        try:
            import matplotlib
            if self.save_plots:
                matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            from matplotlib import patches
        except Exception as e:
            logger.warning("Cannot draw results: matplotlib is not available (%s)" % str(e))
            return
        img = self.image.copy()
        if img.max() == 1:
            img *= 255.
        img = img.astype(np.uint32)
        fig, ax = plt.subplots(1, figsize=(16, 16))
        height, width = img.shape[:2]
        ax.set_ylim(height + 2, -2)
        ax.set_xlim(-2, width + 2)
        ax.axis('off')
        ax.imshow(img)
        for bbox, score, label in zip(self.bboxes_final, self.scores_final, self.labels_final):
            color = self.class_color_map.get(label, (1, 1, 1))
            x1, y1, x2, y2 = (float(v) for v in bbox)
            ax.add_patch(patches.Rectangle((x1, y1), x2 - x1, y2 - y1, linewidth=2, alpha=0.7, linestyle="solid", edgecolor=color,
                                           facecolor='none'))
            if self.draw_class_label_in_caption:
                ax.text(x1, y1 + 8, "{} {:.2f}".format(label, score), color=color, size=20, backgroundcolor="none")
            else:
                ax.text(x1 + (x2 - x1) / 2 - 4, y1 - 1, "{:.2f}".format(score), color="darkturquoise", size=30, backgroundcolor="none")
        logger.info("Write plot to file %s ..." % outfile)
        if self.save_plots:
            fig.savefig(outfile)
            plt.close(fig)
        else:
            plt.show()

    def write_json_results(self, outfile):
        if not self.results:
            logger.warning("Result obj dictionary is empty, nothing to be written...")
            return
        with open(outfile, 'w') as fp:
            json.dump(self.results, fp, indent=2, sort_keys=True)
