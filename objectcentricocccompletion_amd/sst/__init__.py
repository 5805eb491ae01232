from .sst_ops import (build_mlp, get_activation, get_activation_layer, scatter_v2)

__all__ = ['scatter_v2', 'build_mlp', 'get_activation', 'get_activation_layer']
