// Tables shared by the generic (decoder.hip) and the specialised (decoder_fast.hip) decode kernels.
#pragma once
namespace wn {
struct DecCausal { int w, b, ring, cin, cout; };            // float offsets into the arena; b < 0: none
struct DecLayer { int wfg, bfg, wps, bps, ring, d, cd; };
struct DecHead { int w, b, cin, cout; };
struct DecMeta {
    int Q, fwc, ncausal, fw, nlayers, Cr, Cs, nhead, head_act;
    int maxc;          // widest vector that has to sit in LDS
};
size_t decode_fast_pack_floats(int nlayers);
int decode_fast_pack(const WnDecoderDesc* d, float* dst, hipStream_t s);
int decode_fast_launch(const float* P, int nlayers, const float* hbias, const float* E, const DecLayer* layers,
                       float* arena, int* tok_ring, long long n0, int nsteps, int first_token,
                       const double* uniforms, int32_t* out_tokens, float* prob_out, int prob_stride,
                       int apply_softmax, int do_sample, int head_act, bool three_wgs, hipStream_t s);
static constexpr int kDecMaxBatch = 28;                    // utterances per batched launch: 28 x 9 workgroups on 256 CUs
int decode_fast_batch_ok(int nlayers, int n_utt, int nsteps);
int decode_fast_launch_batch(int n_utt, const float* const* P, int nlayers, const float* const* hbias, const float* const* E,
                             const DecLayer* const* layers, float* const* arena, int* const* tok_ring, const long long* n0,
                             int nsteps, const int* first_token, const double* const* uniforms, int32_t* const* out_tokens,
                             float* const* prob_out, int prob_stride, int head_act, bool same_weights, hipStream_t s);
int decode_fast_status(const float* P, int nlayers, hipStream_t s, int* gave_up);
}  // namespace wn
