// piece_spec.hip -- the two-pass pre-pass (piece_filter.hpp) compiled at RUN TIME for one aligner and one read
// length: jit.hpp hands this file, the headers it includes and a generated piece_spec_config.h (the aligner's
// parameters as constexpr objects) to hiprtc with -DATR_SPEC=1 -DATR_SPEC_NW=<words> -DATR_SPEC_RAGGED=<0|1>.
// Never compiled by the Makefile (tools/spec_offline.sh builds it with hipcc for inspection of the ISA).
#include "piece_filter.hpp"

// waves per SIMD the build is bounded for: adapters with more than five body pieces (41 .. 64 bases: eight piece
// accumulators per word, a 96-column window) spill 37 registers at four waves -- they get three
#define ATR_SPEC_WAVES (atr::spec::PP.nb > 5 ? (ATR_PIECE_WAVES(ATR_SPEC_NW) < 3 ? ATR_PIECE_WAVES(ATR_SPEC_NW) : 3) : ATR_PIECE_WAVES(ATR_SPEC_NW))

#ifdef ATR_SPEC_ASCII
// the fused ASCII entry: `planes` is written (the packed batch), the reads come from the ASCII matrix (row stride ATR_SPEC_STRIDE)
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ATR_SPEC_WAVES, 8)))
void atr_piece_spec(const uint8_t *__restrict__ ascii, const int32_t *__restrict__ lens, long long nreads, int max_len,
                    uint4 *__restrict__ out, atr::FastWork wk, uint4 *__restrict__ planes, const atr::PackTableArg tab) {
    atr::piece_filter_body<ATR_SPEC_NW, ATR_SPEC_RAGGED != 0, (atr::spec::PP.window > atr::PIECE_WINDOW ? 3 : 2)>(atr::spec::P, atr::spec::FP, atr::spec::PP, planes, lens, nreads,
                                                             max_len, out, wk, ascii, tab.t);
}
#else
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ATR_SPEC_WAVES, 8)))
void atr_piece_spec(const uint4 *__restrict__ planes, const int32_t *__restrict__ lens, long long nreads, int max_len,
                    uint4 *__restrict__ out, atr::FastWork wk) {
    atr::piece_filter_body<ATR_SPEC_NW, ATR_SPEC_RAGGED != 0, (atr::spec::PP.window > atr::PIECE_WINDOW ? 3 : 2)>(atr::spec::P, atr::spec::FP, atr::spec::PP, planes, lens, nreads,
                                                             max_len, out, wk);
}
#endif
