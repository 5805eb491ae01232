from .sst_ops import (IngroupIndicesFunction, build_mlp, filter_almost_empty, flat2window, flat2window_v2, get_activation,
                      get_activation_layer, get_flat2win_inds, get_flat2win_inds_v2, get_inner_win_inds,
                      get_inner_win_inds_deprecated, get_window_coors, group_rank, make_continuous_inds, scatter_v2,
                      window2flat, window2flat_v2)

__all__ = ['scatter_v2', 'build_mlp', 'get_activation', 'get_activation_layer', 'get_inner_win_inds',
           'make_continuous_inds', 'get_window_coors', 'get_flat2win_inds', 'get_flat2win_inds_v2',
           'flat2window', 'flat2window_v2', 'window2flat', 'window2flat_v2', 'group_rank', 'get_inner_win_inds_deprecated',
           'IngroupIndicesFunction', 'filter_almost_empty']
