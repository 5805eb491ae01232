from . import occ_ops
from .layers import PositionalEncoding, SimpleEncoderLayer, TransformerEncoder
from .occ_base import OccDecoder, PosEncode

__all__ = ['occ_ops', 'PosEncode', 'OccDecoder', 'PositionalEncoding', 'SimpleEncoderLayer', 'TransformerEncoder']
