// K4/K5 MFMA fast paths (gfx950).  Not yet covering any shape: every hook reports RDST_ENOTSUP and
// conv.hip uses the generic functor GEMM.
#include "conv.h"

template <typename T>
int conv_fwd_mfma(const T*, int64_t, int, const float*, const float*, const T*, int64_t, T*, int64_t, const ConvGeom&,
                  float, hipStream_t) { return RDST_ENOTSUP; }
template <typename T>
int conv_dgrad_mfma(const T*, int64_t, int, const float*, const T*, int64_t, T*, int64_t, int, const ConvGeom&, float,
                    hipStream_t) { return RDST_ENOTSUP; }
template <typename T>
int conv_wgrad_mfma(const T*, int64_t, int, const T*, int64_t, float*, float*, const ConvGeom&, float, hipStream_t) {
  return RDST_ENOTSUP;
}
#define INST(T)                                                                                                       \
  template int conv_fwd_mfma<T>(const T*, int64_t, int, const float*, const float*, const T*, int64_t, T*, int64_t,  \
                                const ConvGeom&, float, hipStream_t);                                                \
  template int conv_dgrad_mfma<T>(const T*, int64_t, int, const float*, const T*, int64_t, T*, int64_t, int,         \
                                  const ConvGeom&, float, hipStream_t);                                              \
  template int conv_wgrad_mfma<T>(const T*, int64_t, int, const T*, int64_t, float*, float*, const ConvGeom&, float, \
                                  hipStream_t);
INST(float)
INST(bf16)
