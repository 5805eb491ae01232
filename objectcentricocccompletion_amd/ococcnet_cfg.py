"""The model hyper-parameters of configs/ococc/ococcnet.py:17-181, produced programmatically
(the numbers are data; the reference's file itself is not shipped).  Used by tests, smoke
and the benchmarks on machines where the reference checkout is absent; with the checkout
present ``config.fromfile('<reference>/configs/ococc/ococcnet.py')`` builds the same model."""
from .config import rebuild


def ococcnet_model_cfg(num_blocks=6, d_model=1536, class_names=('Car',)):
    ln = dict(type='LN', eps=1e-3)
    width = d_model // (2 * num_blocks)  # 128: two scatter outputs per block, 6 blocks -> 1536
    bce = lambda red: dict(type='CrossEntropyLoss', use_sigmoid=True, reduction=red, loss_weight=1.0)
    ae = dict(
        type='OccAutoEncoder',
        backbone=dict(type='SIR', num_blocks=num_blocks, in_channels=[15] + [width + 3] * (num_blocks - 1),
                      feat_channels=[[width, width]] * num_blocks, rel_mlp_hidden_dims=[[16, 32]] * num_blocks,
                      with_rel_mlp=True, with_cluster_center=False, with_distance=False, norm_cfg=ln, mode='max',
                      xyz_normalizer=[1, 1, 1], act='gelu', dropout=0, unique_once=True),
        voxel_size=0.2, loss_occ_ae=bce('none'), online_sample_size=-1, balance_sample=True,
        occ_decoder=dict(roi_feature_channels=d_model, occ_mlp=[512, 1024, 1024], use_positional_encoding=True,
                         pos_encode_L=10, norm_pos=True, norm_cfg=ln, act='gelu', occ_dropout=0.1, cls_dim=1,
                         pos_thresh=0.5, use_ln=True),
        with_voxelize_centers=True, compensate_encoder_coors=True)
    head = dict(
        type='OccBBoxHead', num_blocks=num_blocks, in_channels=[24] + [width + 16] * (num_blocks - 1),
        feat_channels=[[width, width]] * num_blocks, rel_mlp_hidden_dims=[[16, 32]] * num_blocks,
        rel_mlp_in_channels=[13] * num_blocks, with_rel_mlp=True, with_cluster_center=False, with_distance=False,
        mode='max', xyz_normalizer=[20, 20, 4], geo_input=True, dropout=0, unique_once=True, occ_ae_head=ae,
        num_classes=len(class_names), roi_feature_channels=d_model, attn_num_head=4, attn_ffn_dim=512,
        attn_dropout=0.1, loss_occ_comp=bce('none'), bbox_coder=dict(type='DeltaXYZWLHRBBoxCoder'),
        occ_label_thresh=0.4, cls_mlp=[512, 512], reg_mlp=[512, 512], latent_mlp=[2048, 2048],
        fusion_mlp=[2048, 2048], act='gelu', norm_cfg=ln,
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=2.0), loss_cls=bce('mean'),
        cls_dropout=0.1, reg_dropout=0.1, latent_dropout=0.1, fusion_dropout=0.1, with_roi_pos_encoding=True,
        roi_pos_enc_mlp=[512, 512], num_enc_layers=3, fixed_ae=False, fused_mode='concat', rcnn_trans=False)
    model = dict(
        type='TrackletDetectorOCC',
        roi_head=dict(
            type='TrackletRoIHeadOCC', num_classes=len(class_names), general_cfg=dict(with_roi_scores=True),
            history_only=True,
            roi_extractor=dict(type='TrackletPointRoIExtractor', extra_wlh=[0.5, 0.5, 0.5], max_inbox_point=4096,
                               max_all_point=(300000, 600000), debug=False, combined=False),
            bbox_head=head, pretrained=None),
        train_cfg=dict(pre_voxelization_size=None, assigner=dict(type='TrackletAssigner'), hack_sampler_bug=True,
                       cls_pos_thr=(0.8,), cls_neg_thr=(0.2,), sync_reg_avg_factor=True, sync_cls_avg_factor=True,
                       corner_loss_only_car=True, class_names=list(class_names),
                       rcnn_code_weights=[2.0, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0], fixed_length=True,
                       num_occ_per_tracklet=-1, random_shift_frame_inds=True, keep_frame_inds=False,
                       residual_loss=False, contrastive_loss=False, no_loss_for_outside=False,
                       no_loss_for_observed_feats=False, contrastive_loss_weight=1.0),
        test_cfg=dict(batch_inference=True, test_occ_iou=True, iou_chunk_size=10, ignore_outside_occ=True,
                      test_baseline=False))
    return rebuild(model)


def ococcnet_train_pipeline(reg_len=32, occ_voxel_size=0.2, class_names=('Car',)):
    """train_pipeline of configs/ococc/ococcnet.py:183-262 (built by pipelines.Compose / dataset.py)."""
    return [
        dict(type='LoadTrackletPoints', load_dim=6, use_dim=5, max_points=1024, debug=False),
        dict(type='LoadTrackletAnnotations'),
        dict(type='LoadAnnotationsOcc', compute_score=False),
        dict(type='RandomSampleOccPoints', num_sample_points=512, pos_sample_weight=0.5, voxel_size=occ_voxel_size,
             use_unknown=False, use_potential=False, balance_sample=True, weighted_sample=True),
        dict(type='TrackletRegularization', reg_len=reg_len),
        dict(type='TrackletPoseTransform', concat=False),
        dict(type='TrackletNoise', center_noise_cfg=dict(max_noise=[0.2, 0.2, 0.1], consistent=False),
             size_noise_cfg=dict(max_noise=[0.2, 0.2, 0.1], consistent=False),
             yaw_noise_cfg=dict(max_noise=0.2, consistent=False)),
        dict(type='PointDecoration', properties=['yaw', 'size', 'score'], concat=True),
        dict(type='TrackletRandomFlip', flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
        dict(type='TrackletGlobalRotScaleTrans', rot_range=[-0.78539816, 0.78539816], scale_ratio_range=[0.95, 1.05],
             translation_std=[0, 0, 0.2]),
        dict(type='PointsRangeFilter', point_cloud_range=[-204.7, -204.7, -3.99, 204.7, 204.7, 7.99]),
        dict(type='PointShuffle'),
        dict(type='TrackletOccFormatBundle', class_names=list(class_names)),
        dict(type='Collect3D', keys=['points', 'pts_frame_inds', 'tracklet', 'gt_tracklet_candidates', 'occ_labels',
                                     'occ_labels_scores']),
    ]
