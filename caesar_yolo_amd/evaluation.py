"""Per-frame detect object: host mirror of the reference's `Analyzer` (caesar_yolo/evaluation.py:38-556) on the HIP path.

`Analyzer(model, config).predict(image, image_id, header, xmin, ymin)` keeps the reference's contract: 0 / -1 return,
results in `.bboxes_final / .scores_final / .class_ids_final / .labels_final / .results`; the work between the image and
the merged boxes (3-channel cube, preprocessing, rejection checks, model call, score filter + IoU graph merge:
evaluation.py:146-346) is one `cy_detect_tiles` call with the frame as a single tile.  Plotting and DS9 output are out
of scope."""
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
        if image.ndim != 2:
            logger.error("The HIP path takes single-channel 2-D frames (the 3-channel cube is built on device)")
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
        st = int(status[0])
        if st == 1:
            logger.warning("Input image is None, no prediction made.")
            return -1
        if st == 2:
            logger.warning("Input image pixels have the same value in one of the first rows, no prediction made.")
            return -1
        dd = d[0, :int(cnt[0])].cpu().numpy()
        self.bboxes_final = [dd[i, :4].copy() for i in range(dd.shape[0])]
        self.scores_final = [dd[i, 4] for i in range(dd.shape[0])]
        self.class_ids_final = [int(dd[i, 5]) for i in range(dd.shape[0])]
        self.labels_final = [self.class_names[c] for c in self.class_ids_final]
        self.results = {"image_id": self.image_id,
                        "objs": objs_from_detections(dd, self.class_names, nx, ny, xmin, ymin, self.obj_name_tag)}
        if self.write_to_json:
            self.write_json_results(self.outfile_json or ('out_' + str(self.image_id) + '.json'))
        if self.write_to_ds9:                                     # caesar_yolo/evaluation.py:228-234
            self.write_ds9_regions(self.outfile_ds9 or ('out_' + str(self.image_id) + '.reg'))
        return 0

    def write_ds9_regions(self, outfile):
        from . import utils
        utils.write_ds9_regions(outfile, self.results.get("objs", []), merged_tag=False)

    def write_json_results(self, outfile):
        if not self.results:
            logger.warning("Result obj dictionary is empty, nothing to be written...")
            return
        with open(outfile, 'w') as fp:
            json.dump(self.results, fp, indent=2, sort_keys=True)
