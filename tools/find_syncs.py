"""Host synchronisations of one --workload ococcnet training step, by call site (torch sync-debug warnings)."""
import collections
import os

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')  # before the HIP runtime loads: objectcentricocccompletion_amd/graph.py
import sys
import traceback
import warnings

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
params = [p for p in model.parameters() if p.requires_grad]
opt = AdamW(params, lr=1e-6)
batch = synthetic_training_batch(int(os.environ.get('TRACKLETS', '4')), 32, pts_per_frame=64, occ_queries=512, seed=0, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    losses = model(return_loss=True, **batch)
    total = losses['loss_rcnn_cls'] + losses['loss_rcnn_bbox'] + losses['loss_rcnn_occ'].mean()
    total.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
sites = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def show(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' not in str(message):
        return
    for fr in reversed(traceback.extract_stack()):
        if fr.filename.startswith(root) and 'find_syncs' not in fr.filename:
            sites[f'{os.path.relpath(fr.filename, root)}:{fr.lineno} {fr.line}'] += 1
            return
    sites[f'{filename}:{lineno}'] += 1


warnings.showwarning = show
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step()
torch.cuda.set_sync_debug_mode('default')
for k, v in sites.most_common():
    print(v, k)
print('total', sum(sites.values()))
