// MFMA implicit-GEMM conv instantiations for 5x5x5 stride 1 (see conv3d_mfma.h)
#include "conv3d_mfma.h"

CFUN_MFMA_DEFINE(k555s1, 5, 5, 5, 1)
