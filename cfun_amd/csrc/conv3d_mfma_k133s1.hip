// MFMA implicit-GEMM conv instantiations for 1x3x3 stride 1 (see conv3d_mfma.h)
#include "conv3d_mfma.h"

CFUN_MFMA_DEFINE(k133s1, 1, 3, 3, 1)
