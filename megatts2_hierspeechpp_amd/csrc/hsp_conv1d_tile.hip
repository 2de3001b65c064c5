// One tile shape of the fused conv kernel (hsp_conv1d_mfma_kernel.h) per translation unit: the Makefile
// compiles this file once per shape with -DHSP_TILE=<shape>, so the shapes build in parallel.
// Each shape carries only the (epilogue kind, activation prologue) combinations the dispatcher
// (hsp_conv1d_mfma.hip) routes to it.
#include "hsp_conv1d_mfma_kernel.h"

#ifndef HSP_TILE
#error "compile with -DHSP_TILE=<M128|M64|M32|M64P|M32P|S64|S64G|S32|S64W|S64GW|S64G2>"
#endif

namespace {
using namespace hspconv;
using T = HSP_TILE;

template <class A, class B>
constexpr bool same = std::is_same<A, B>::value;

template <int EPI, bool ACT>
constexpr bool supported() {
  if (same<T, S64G> || same<T, S64GW> || same<T, S64G2>) return EPI == HSP_EPI_GATE && !ACT;
  if (same<T, M64> || same<T, M32>) return ACT && (EPI == HSP_EPI_INIT || EPI == HSP_EPI_GEN);  // activation shapes
  if (ACT) return (same<T, M128> || same<T, S64> || same<T, S64W> || same<T, S32>) && (EPI == HSP_EPI_INIT || EPI == HSP_EPI_GEN);
  if (EPI == HSP_EPI_GATE) return same<T, M128>;
  if (EPI == HSP_EPI_SHUF) return true;  // every plain shape (ConvTranspose of a single short utterance: S64 / S32)
  return true;  // INIT / VEC / GEN on every plain shape
}

template <int EPI, bool ACT>
int try_one(const hsp_conv1d_args& a, hipStream_t s, int32_t* plan_out) {
  if constexpr (supported<EPI, ACT>()) return launch_one<T, EPI, ACT>(a, s, plan_out);
  else return HSP_EINVAL;
}

template <int EPI>
int by_act(const hsp_conv1d_args& a, bool act, hipStream_t s, int32_t* plan_out) {
  return act ? try_one<EPI, true>(a, s, plan_out) : try_one<EPI, false>(a, s, plan_out);
}
}  // namespace

#define HSP_CAT_(a, b) a##b
#define HSP_CAT(a, b) HSP_CAT_(a, b)

int HSP_CAT(hsp_conv_tile_, HSP_TILE)(const hsp_conv1d_args& a, int epi, bool act, hipStream_t s, int32_t* plan_out) {
  switch (epi) {
    case HSP_EPI_INIT: return by_act<HSP_EPI_INIT>(a, act, s, plan_out);
    case HSP_EPI_VEC: return by_act<HSP_EPI_VEC>(a, act, s, plan_out);
    case HSP_EPI_GATE: return by_act<HSP_EPI_GATE>(a, act, s, plan_out);
    case HSP_EPI_SHUF: return by_act<HSP_EPI_SHUF>(a, act, s, plan_out);
    case HSP_EPI_GEN: return by_act<HSP_EPI_GEN>(a, act, s, plan_out);
    default: return HSP_EINVAL;
  }
}
