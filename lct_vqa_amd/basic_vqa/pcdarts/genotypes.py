"""Search space of the reference (basic_vqa/pcdarts/genotypes.py:3-14).  The reference file also lists
named genotypes of published architectures; they are not used by the search path and are not repeated."""
from collections import namedtuple

Genotype = namedtuple('Genotype', 'normal normal_concat reduce reduce_concat')

PRIMITIVES = [
    'none',
    'max_pool_3x3',
    'avg_pool_3x3',
    'skip_connect',
    'sep_conv_3x3',
    'sep_conv_5x5',
    'dil_conv_3x3',
    'dil_conv_5x5'
]
