// MFMA implicit-GEMM conv instantiations for 2x2x2 stride 1 -- the parity-folded data gradient of the
// stride-2 3x3x3 convs (see cfun_conv3d_bwd_data in conv3d.hip and conv3d_mfma.h)
#include "conv3d_mfma.h"

CFUN_MFMA_DEFINE(k222s1, 2, 2, 2, 1)
