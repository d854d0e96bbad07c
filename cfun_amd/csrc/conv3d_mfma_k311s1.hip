// MFMA implicit-GEMM conv instantiations for 3x1x1 stride 1 (see conv3d_mfma.h)
#include "conv3d_mfma.h"

CFUN_MFMA_DEFINE(k311s1, 3, 1, 1, 1)
