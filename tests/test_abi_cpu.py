"""CPU suite, part 2: the C ABI library loads and exports every symbol include/ococc_hip.h
declares; host-side mirrors keep the reference's names, signatures and module trees; ops
refuse to run without a device (no CPU fallback)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'ococc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ococc_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from objectcentricocccompletion_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in ococc_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in _lib.py'
    assert _lib.lib.ococc_arch() == b'gfx950'
    assert _lib.lib.ococc_version() >= 100


def test_workspace_queries_are_host_arithmetic():
    from objectcentricocccompletion_amd import _lib as L
    assert L.lib.ococc_grid_unique_workspace_bytes(4, L.i4([64, 40, 40, 40])) > 0
    assert L.lib.ococc_grid_unique_workspace_bytes(5, L.i4([1, 1, 1, 1])) == -1
    assert L.lib.ococc_subm_rulebook_workspace_bytes(1000, 2, L.i3([40, 40, 40]), L.i3([3, 3, 3])) > 0
    assert L.lib.ococc_subm_rulebook_workspace_bytes(1000, 2, L.i3([40, 40, 40]), L.i3([2, 3, 3])) == -1
    assert L.lib.ococc_sparse_conv_wgrad_workspace_bytes(27, 1000, 64, 128) >= 27 * 64 * 128 * 4
    assert L.lib.ococc_layernorm_act_bwd_workspace_bytes(1000, 128) > 0


def test_argument_errors_are_reported_not_thrown():
    from objectcentricocccompletion_amd import _lib as L
    rc = L.lib.ococc_dynamic_voxelize_f32(None, 10, 2, L.f3([.2, .2, .2]), L.f6([0, 0, 0, 1, 1, 1]), None, None)
    assert rc == -1 and b'num_features' in L.lib.ococc_last_error()
    with pytest.raises(L.OcoccError):
        L.check(rc, 'dynamic_voxelize')


def test_ops_refuse_cpu_tensors():
    from objectcentricocccompletion_amd import _lib as L
    from objectcentricocccompletion_amd.voxel import dynamic_scatter, voxelization
    from objectcentricocccompletion_amd.norm import layer_norm_act
    with pytest.raises(L.OcoccError):
        voxelization(torch.zeros(4, 3), [.2, .2, .2], [0, 0, 0, 1, 1, 1], -1, -1)
    with pytest.raises(L.OcoccError):
        dynamic_scatter(torch.zeros(4, 3), torch.zeros(4, 3, dtype=torch.int32), 'max')
    with pytest.raises(L.OcoccError):
        layer_norm_act(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_reference_signatures_are_kept():
    from objectcentricocccompletion_amd.spconv import ops
    from objectcentricocccompletion_amd.voxel import voxelize
    # mmdet3d/ops/spconv/ops.py:46-56,109-116,142-149
    assert list(inspect.signature(ops.get_indice_pairs).parameters)[:11] == [
        'indices', 'batch_size', 'spatial_shape', 'ksize', 'stride', 'padding', 'dilation',
        'out_padding', 'subm', 'transpose', 'grid']
    assert list(inspect.signature(ops.indice_conv).parameters)[:7] == [
        'features', 'filters', 'indice_pairs', 'indice_pair_num', 'num_activate_out', 'inverse', 'subm']
    assert list(inspect.signature(ops.indice_conv_backward).parameters)[:7] == [
        'features', 'filters', 'out_bp', 'indice_pairs', 'indice_pair_num', 'inverse', 'subm']
    # mmdet3d/ops/voxel/voxelize.py:13-18
    assert list(inspect.signature(voxelize._Voxelization.forward).parameters)[1:] == [
        'points', 'voxel_size', 'coors_range', 'max_points', 'max_voxels']
    assert ops.get_conv_output_size([41, 1600, 1408], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]) == [21, 800, 704]


def test_sparse_convmodule_tree_and_state_dict_names():
    from objectcentricocccompletion_amd.sparse_block import SparseBasicBlock, make_sparse_convmodule
    from objectcentricocccompletion_amd.spconv import SparseSequential, SubMConv3d
    m = make_sparse_convmodule(16, 32, 3, 'subm1', padding=1, conv_type='SubMConv3d', act_type='gelu',
                               norm_cfg=dict(type='LN', eps=1e-3))
    assert isinstance(m, SparseSequential) and isinstance(m[0], SubMConv3d)
    sd = m.state_dict()
    assert list(sd) == ['0.weight', '1.weight', '1.bias']
    assert tuple(sd['0.weight'].shape) == (3, 3, 3, 16, 32)  # (kD,kH,kW,Cin,Cout), conv.py:98-99
    assert m[1].eps == 1e-3 and m[1].fused_act == 'gelu'
    blk = SparseBasicBlock(16, 16, conv_cfg=dict(type='SubMConv3d', indice_key='k'), norm_cfg=dict(type='BN1d'))
    assert {'conv1.weight', 'bn1.weight', 'conv2.weight', 'bn2.running_mean'} <= set(blk.state_dict())
    from objectcentricocccompletion_amd.sparse_block import SparseBottleneck
    bot = SparseBottleneck(64, 16, conv_cfg=dict(type='SubMConv3d', indice_key='k'), norm_cfg=dict(type='BN1d'))
    sd = bot.state_dict()   # mmdet's Bottleneck names (sparse_block.py:22-78): 1x1 -> 3x3 -> 1x1 onto planes * 4
    assert {'conv1.weight', 'bn1.weight', 'conv2.weight', 'bn2.weight', 'conv3.weight', 'bn3.running_var'} <= set(sd)
    assert tuple(sd['conv1.weight'].shape) == (1, 1, 1, 64, 16) and tuple(sd['conv3.weight'].shape) == (1, 1, 1, 16, 64)
    from objectcentricocccompletion_amd.sparse_block import AdaptiveSparseBasicBlock
    ada = AdaptiveSparseBasicBlock(16, 32, stride=2, conv_cfg=dict(type='SubMConv3d', indice_key='a'), norm_cfg=dict(type='BN1d'))
    sd = ada.state_dict()   # sparse_block.py:146-213: a strided SparseConv3d named '<key>.adaptive' in front of the block
    assert {'ada_conv.weight', 'ada_norm.running_mean', 'conv1.weight', 'bn2.weight'} <= set(sd)
    assert tuple(sd['ada_conv.weight'].shape) == (2, 2, 2, 16, 32) and ada.ada_conv.indice_key == 'a.adaptive'
    assert not hasattr(AdaptiveSparseBasicBlock(32, 32, conv_cfg=dict(type='SubMConv3d', indice_key='a'), norm_cfg=dict(type='BN1d')), 'ada_conv')


def test_spconv_container_helpers():
    """is_sparse_conv / _mean_update / RemoveGrid of the reference's spconv/modules.py (:27-43, :197-202)"""
    import torch
    from objectcentricocccompletion_amd.spconv import RemoveGrid, SparseConvTensor, SubMConv3d
    from objectcentricocccompletion_amd.spconv.modules import _mean_update, is_sparse_conv, is_spconv_module
    conv = SubMConv3d(4, 8, 3, indice_key='k')
    assert is_sparse_conv(conv) and is_spconv_module(conv) and not is_sparse_conv(torch.nn.ReLU())
    assert _mean_update(4.0, 1.0, 2) == 2.0 and _mean_update([3.0, 6.0], [0.0, 0.0], 2) == [1.0, 2.0]
    t = SparseConvTensor(torch.zeros(1, 4), torch.zeros(1, 4, dtype=torch.int32), [2, 2, 2], 1, grid=torch.zeros(1))
    assert RemoveGrid()(t).grid is None
