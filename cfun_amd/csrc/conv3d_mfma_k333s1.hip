// MFMA implicit-GEMM conv instantiations for 3x3x3 stride 1 (see conv3d_mfma.h)
#include "conv3d_mfma.h"

CFUN_MFMA_DEFINE(k333s1, 3, 3, 3, 1)
