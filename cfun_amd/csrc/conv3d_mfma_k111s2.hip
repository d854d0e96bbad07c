// MFMA implicit-GEMM conv instantiations for 1x1x1 stride 2 (see conv3d_mfma.h)
#include "conv3d_mfma.h"

CFUN_MFMA_DEFINE(k111s2, 1, 1, 1, 2)
