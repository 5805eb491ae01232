# OcOccNet on MI355X: the model dict of the reference's configs/ococc/ococcnet.py (built field by field in
# objectcentricocccompletion_amd/ococcnet_cfg.py, 66 553 173 parameters, same state-dict names), its train pipeline,
# optimizer and batch size.  Usage: python tools/train.py configs/ococcnet_mi355x.py [--data-root DIR]
from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg, ococcnet_train_pipeline

model = ococcnet_model_cfg()
train_pipeline = ococcnet_train_pipeline()
data = dict(samples_per_gpu=4, workers_per_gpu=4)
optimizer = dict(type='AdamW', lr=1e-6, weight_decay=0.01)          # configs/ococc/ococcnet.py: lr, AdamW
optimizer_config = dict(grad_clip=dict(max_norm=10, norm_type=2))
