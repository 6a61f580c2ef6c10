// The MixedOp edge kernels with the nine pre-BN tensors stored as bf16 (fp32 arithmetic and statistics): the same source compiled a second
// time; entry points lctvqa_mixed_edge_fwd_bf16pre / lctvqa_mixed_edge_bwd_bf16pre.  See the head of mixed_edge.hip (pre_t).
#define EDGE_PRE_BF16 1
#include "mixed_edge.hip"
