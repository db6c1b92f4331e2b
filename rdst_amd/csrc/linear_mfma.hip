// K3 MFMA fast paths (gfx950).  Not yet covering any shape: every hook reports RDST_ENOTSUP and
// linear.hip uses the generic functor GEMM.
#include "linear.h"

template <typename T>
int linear_fwd_mfma(const T*, int64_t, const float*, const float*, int, const float*, const float*, const T*, int64_t,
                    T*, int64_t, float*, int64_t, int, int, float, hipStream_t) { return RDST_ENOTSUP; }
template <typename T>
int linear_dgrad_mfma(const T*, int64_t, bool, int, const float*, const T*, int64_t, T*, int64_t, int, float*, int64_t,
                      int, int, float, hipStream_t) { return RDST_ENOTSUP; }
template <typename T>
int linear_wgrad_mfma(const T*, int64_t, const float*, const float*, const float*, int, const T*, int64_t, float*,
                      float*, int64_t, int, int, float, hipStream_t) { return RDST_ENOTSUP; }

#define INST(T)                                                                                                        \
  template int linear_fwd_mfma<T>(const T*, int64_t, const float*, const float*, int, const float*, const float*,     \
                                  const T*, int64_t, T*, int64_t, float*, int64_t, int, int, float, hipStream_t);     \
  template int linear_dgrad_mfma<T>(const T*, int64_t, bool, int, const float*, const T*, int64_t, T*, int64_t, int,  \
                                    float*, int64_t, int, int, float, hipStream_t);                                   \
  template int linear_wgrad_mfma<T>(const T*, int64_t, const float*, const float*, const float*, int, const T*,       \
                                    int64_t, float*, float*, int64_t, int, int, float, hipStream_t);
INST(float)
INST(bf16)
