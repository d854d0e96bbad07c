// MFMA implicit-GEMM conv instantiations for 3x3x3 stride 2 (see conv3d_mfma.h)
#include "conv3d_mfma.h"

CFUN_MFMA_DEFINE(k333s2, 3, 3, 3, 2)
