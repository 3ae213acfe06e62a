"""MI355X-native tiled-YOLO detect path for caesar-yolo."""
import os

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two BUSY streams on one queue
# serialise.  A detector context owns four streams (caller's, preprocessing, post-processing, second forward); a collective library
# in the process (RCCL: one stream per communicator) makes five, and its stream -- created first when the process group is
# initialised eagerly -- pushes two of ours onto one queue: +5-6 % per pass, measured with a one-rank RCCL group (DESIGN.md section 5,
# tools/ab_streams.sh).  Eight queues cost nothing without it and remove the effect with it.  Only effective when set before the
# first HIP call of the process: bench.py and scripts/run.py set it before they touch torch.cuda as well.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
