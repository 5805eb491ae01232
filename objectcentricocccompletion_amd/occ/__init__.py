from . import occ_ops
from .layers import (PositionalEncoding, SimpleDecoderLayer, SimpleEncoderLayer, TransformerDecoder,
                     TransformerEncoder)
from .occ_base import OccDecoder, PosEncode

__all__ = ['occ_ops', 'PosEncode', 'OccDecoder', 'PositionalEncoding', 'SimpleEncoderLayer',
           'TransformerEncoder', 'SimpleDecoderLayer', 'TransformerDecoder']
