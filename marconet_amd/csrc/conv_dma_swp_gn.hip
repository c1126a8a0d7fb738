// The software-pipelined 256x256 fp16+8 tile WITH the GroupNorm-sum block in its epilogue (conv_dma_kernel<..., SWP, SGN = true>), in a translation unit of its
// own so that it can be compiled with `-mllvm -greedy-reverse-local-assignment=1` (marconet_amd/csrc/build.sh): under hipcc's default assignment order that block puts
// scratch reloads into the tile's slab loop, under the reverse order the tile is clean — and no other kernel's register allocation is touched by the flag.
// Same MFMA sequence per output as every fp16+8 tile: the same bytes (tests/test_mx_gpu.py::test_groupnorm_partial_sums_from_the_conv_epilogue).
#define MNET_DMA_SWP_GN_TU 1
#include "conv_igemm_dma.hip"
