"""cProfile of the host side of one --workload ococcnet training step (cumulative time by function)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectcentricocccompletion_amd import heads, point_pool, roi_head  # noqa: F401
from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg
from objectcentricocccompletion_amd.optim import AdamW
from objectcentricocccompletion_amd.registry import DETECTORS
from objectcentricocccompletion_amd.synthetic import synthetic_training_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = ococcnet_model_cfg()
cfg['train_cfg']['random_shift_frame_inds'] = False
model = DETECTORS.build(cfg).to(dev).train()
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-6)
batch = synthetic_training_batch(4, 32, pts_per_frame=64, occ_queries=512, seed=0, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    losses = model(return_loss=True, **batch)
    total = losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()
    total.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
