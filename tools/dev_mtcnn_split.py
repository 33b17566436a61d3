"""Developer aid: cProfile of MTCNN stage 1 on a 1280x720 frame (where does the host time go?)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from facenet_amd.detectors import mtcnn as gm
from oracle import mtcnn_oracle as mo
from tools.bench_mtcnn import frame

det = gm.MTCNN(weights=mo.random_weights(0, face_bias=(float(sys.argv[1]) if len(sys.argv) > 1 else -0.3, 1.0, 1.0)))
f = torch.from_numpy(frame(720, 1280, 0)).cuda()
for _ in range(3):
    det.detect_boxes(f)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    det.detect_boxes(f)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
