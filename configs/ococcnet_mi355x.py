# OcOccNet on MI355X: the model dict of the reference's configs/ococc/ococcnet.py (built field by field in
# objectcentricocccompletion_amd/ococcnet_cfg.py, 66 553 173 parameters, same state-dict names), its train pipeline,
# optimizer and batch size.  Usage: python tools/train.py configs/ococcnet_mi355x.py [--data-root DIR]
from objectcentricocccompletion_amd.ococcnet_cfg import ococcnet_model_cfg, ococcnet_train_pipeline

model = ococcnet_model_cfg()
train_pipeline = ococcnet_train_pipeline()
data = dict(samples_per_gpu=4, workers_per_gpu=4)
# configs/_base_/schedules/cosine_2x.py:2-15 merged with configs/ococc/ococcnet.py:468-470 (lr override)
optimizer = dict(type='AdamW', lr=1e-6, betas=(0.9, 0.999), weight_decay=0.05,
                 paramwise_cfg=dict(custom_keys={'norm': dict(decay_mult=0.)}))
lr_config = dict(policy='cyclic', target_ratio=(100, 1e-3), cyclic_times=1, step_ratio_up=0.1)
optimizer_config = dict(grad_clip=dict(max_norm=10, norm_type=2))
