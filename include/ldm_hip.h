/* C ABI of libldm_hip.so: the MI355X (gfx950) implementation of the latent-diffusion sampling hot
 * path of ragu812/diffusionModel -- UNet forward, timestep embedding, DDPM/DDIM scheduler step and
 * the sampling loop.  Plain pointers and sizes only; every `*_dev` pointer is device memory of the
 * current HIP device, `stream` is a hipStream_t (NULL = default stream).  Every function returns
 * LDM_OK or an error code; ldm_last_error() gives the text for the calling thread.
 *
 * Tensor conventions at this boundary are the reference's: activations NCHW f32, Conv2d weights
 * [Cout,Cin,kh,kw], ConvTranspose2d weights [Cin,Cout,kh,kw], Linear weights [d_in,d_out]
 * (burn's layouts).  Internally activations are NHWC bf16 or f32.
 *
 * Deviations from the reference as written (it cannot run as written):
 *  - down-sampling uses the stride-2 convolutions down2/down3; pool1/pool2 (src/main.rs:432,443,
 *    502,506) are skipped, which is the only shape-consistent reading of Unet::forward;
 *  - inference semantics: BatchNorm uses running statistics (folded into the preceding conv,
 *    eps 1e-5), Dropout is the identity;
 *  - the per-step NaN probes of denoising_process (src/main.rs:706-710,725-729,759-763) are host
 *    synchronisations and are not reproduced; the probes after the loop (:780-784, :788-792) and those of
 *    DiffusionModel::forward (:647-651, 657-662, 672-676) are offered as ldm_latent_is_finite, which the caller
 *    runs after the loop (the Python surface: check_finite=True).
 *
 * Threads and streams: a module or sampler owns ONE set of activation buffers (a module's grows on demand, which
 * synchronises the device and reallocates).  Calls on the same module or sampler must not overlap: one stream at a
 * time, one host thread at a time.  Different modules and samplers are independent.
 *
 * Environment variables read by the library (developer switches; the defaults are what is measured and tested):
 *   LDM_DENSE4=0           keep the tiled convolution kernels at 4x4 latents instead of the whole-image matrices.
 *                          ldm_unet_forward reads it at every call, ldm_sampler_create once (the sampler keeps
 *                          its choice).
 *   LDM_SKINNY3_MAXWG=n    3x3 convolutions with at most n skinny workgroups take the skinny kernel (default: 1024
 *                          in f32, 0 in bf16); read once per process.
 *   LDM_RESFUSE=0          run a ResidualBlock of the 64-channel level as its two convolution launches instead of the
 *                          one fused kernel (same bits; comparison runs and tests); read once per process.
 *   LDM_DDIMCHAIN=0        bf16 DDIM sampling: keep out_conv + DDIM update and the next step's down1 as two launches instead
 *                          of the one chained kernel (same bits; comparison runs and tests); read by ldm_sampler_create.
 *   LDM_ATTN_FOLD=0        run the bf16 Attention block (256 / 512 channels) with the reference's four Linears instead of
 *                          the projections folded at pack time (same mathematics, other roundings; comparison runs and
 *                          tests); read at every call (a captured sampler graph keeps the choice made at capture).
 */
#ifndef LDM_HIP_H
#define LDM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum { LDM_OK = 0, LDM_ERR_INVALID = 1, LDM_ERR_HIP = 2, LDM_ERR_STATE = 3 };
enum { LDM_F32 = 0, LDM_BF16 = 1 };                 /* arithmetic type of weights and activations */
enum { LDM_DDPM = 0, LDM_DDIM = 1 };
enum { LDM_MOD_UNET = 0, LDM_MOD_RESBLOCK = 1, LDM_MOD_ATTENTION = 2, LDM_MOD_TIME_ADDITION = 3, LDM_MOD_VAE_DECODER = 4 };

typedef struct ldm_module ldm_module;
typedef struct ldm_sampler ldm_sampler;

const char* ldm_last_error(void);
int ldm_version(void);

/* ---- noise schedule (host, f32, operation order of the reference) ------------------------- */
/* DiffusionModel::get_betas, src/main.rs:553-567: linear 1e-4 -> 0.02 clamped to [1e-6, 0.999]. */
int ldm_get_betas(int num_time, float* out_host);
/* DiffusionModel::get_alphas, src/main.rs:569-583: returns the running product alpha-bar. */
int ldm_get_alphas(int num_time, float* out_host);
/* Scalars of the reverse step, src/main.rs:712-747; row i = {sqrt(abar_i), sqrt(1-abar_i), coef1,
 * coef2, sigma}; row 0 = {.., .., 1, 0, 0} (the reference sets z = x0 there, :765). */
int ldm_ddpm_coefficients(int num_time, float* out_host /* [num_time][5] */);
/* DDIM (eta = 0) has no counterpart in the reference.  Timesteps t_k = k * (num_time / steps),
 * visited from k = steps-1 down to 0; row k (visit order) = {sqrt(abar_t), sqrt(1-abar_t),
 * sqrt(abar_prev), sqrt(1-abar_prev), 0}, abar_prev = abar[t - num_time/steps] or 1 below zero. */
int ldm_ddim_timesteps(int num_time, int steps, int* out_host /* [steps] */);
int ldm_ddim_coefficients(int num_time, int steps, float* out_host /* [steps][5] */);

/* ---- sin_time_addition, src/main.rs:309-335 ------------------------------------------------- */
/* out[b, i] = sin(t_b f_i), out[b, dim/2 + i] = cos(t_b f_i), f_i = exp(-(i * 4 ln10 / (dim/2))). */
int ldm_sin_time_addition(const int* time_dev, int batch, int dim, float* out_dev, void* stream);

/* ---- modules ------------------------------------------------------------------------------ */
/* kind LDM_MOD_UNET:          Unet::new(in_channels, time_embed_dim), src/main.rs:417-494;
 *                             channels3 = {64,128,256} when NULL; scaled_attention adds one
 *                             Attention(channels3[1]) after up_res1 (named "up_atten").
 * kind LDM_MOD_RESBLOCK:      ResidualBlock::new(channels = in_channels, time_embed_dim), :346-370
 * kind LDM_MOD_ATTENTION:     Attention::new(channels = in_channels), :247-255
 * kind LDM_MOD_TIME_ADDITION: TimeAddition::new(dim = in_channels), :285-300
 * Parameters are set by their Rust field path ("down_res1.conv1.weight", "mid_atten.question.bias",
 * "down_res1.bn1.running_var", "time_emb.linear1.weight"; for the single-block kinds without the
 * block prefix: "conv1.weight", "question.weight", "linear1.weight") from host f32 arrays, then
 * ldm_module_finalize folds BatchNorm, packs for the kernels and uploads. */
int ldm_module_create(ldm_module** out, int kind, int dtype, int in_channels, int time_embed_dim,
                      const int* channels3, int scaled_attention);
int ldm_module_set_param(ldm_module* m, const char* name, const float* host_data, long numel);
int ldm_module_finalize(ldm_module* m);
void ldm_module_destroy(ldm_module* m);

/* Unet::forward(x, time), src/main.rs:496-526.  x [B,in_channels,H,W], time_emb [B,time_embed_dim]
 * (the sinusoidal embedding), eps [B,in_channels,H,W]; H and W multiples of 4. */
int ldm_unet_forward(ldm_module* m, const float* x_dev, const float* time_emb_dev, float* eps_dev,
                     int B, int H, int W, void* stream);
/* ResidualBlock::forward(x, t_emb), src/main.rs:372-387.  x/out [B,channels,H,W], t_emb [B,time_embed_dim]. */
int ldm_resblock_forward(ldm_module* m, const float* x_dev, const float* t_emb_dev, float* out_dev,
                         int B, int H, int W, void* stream);
/* Attention::forward(x), src/main.rs:256-276.  x/out [B,channels,H,W]. */
int ldm_attention_forward(ldm_module* m, const float* x_dev, float* out_dev, int B, int H, int W, void* stream);
/* TimeAddition::forward(t), src/main.rs:302-306.  t/out [B,dim]. */
int ldm_time_addition_forward(ldm_module* m, const float* t_dev, float* out_dev, int B, void* stream);

/* ---- training-mode forward: the noise prediction as DiffusionModel::forward computes it -------- */
/* DiffusionModel::forward (src/main.rs:635-679) runs Unet::forward on the autodiff backend (:1318), where burn's
 * BatchNorm (bn1 / bn2 of every ResidualBlock, :340-341, 356, 366, 374, 385) normalises with the statistics of the
 * batch -- per channel the mean and the biased variance over (B, H, W), eps 1e-5 -- and mid_drop (:403, 458, 513)
 * zeroes each element with probability dropout_p (the reference's 0.5) and scales the rest by 1 / (1 - dropout_p).
 * Forward value only: no gradients, and the modules' running statistics are not touched.
 * time_emb [B, time_embed_dim]: one sinusoidal row per image (the reference draws a timestep per image, :639-653).
 * The dropout mask comes from the library's counter-based generator: element (b, c, y, x) of the
 * [B, C, H/4, W/4] tensor is kept when word (c & 3) of Philox-4x32-10 with counter
 * {((y W/4 + x) C + c) / 4, 0xFFFFFFFE, b, 0x4C444D32} and key dropout_seed gives ((word >> 8) + 0.5) 2^-24 < 1 - p.
 * bn_mean_dev / bn_var_dev: optional [2 * (sum of the seven blocks' channels)] f32, per block (down_res1,
 * down_res2, down_res3, mid_res1, mid_res2, up_res1, up_res2) bn1's channels then bn2's: the batch statistics, from
 * which the caller can make burn's running-average update (momentum 0.1).
 * The first training-mode call on a module packs the unfolded convolutions (host work and device allocations). */
int ldm_unet_forward_train(ldm_module* m, const float* x_dev, const float* time_emb_dev, float* eps_dev, int B, int H, int W,
                           float dropout_p, unsigned long long dropout_seed, float* bn_mean_dev, float* bn_var_dev,
                           void* stream);
/* ResidualBlock::forward (src/main.rs:372-387) with bn1 / bn2 on the statistics of the batch.
 * bn_mean_dev / bn_var_dev: optional [2 * channels] (bn1 then bn2). */
int ldm_resblock_forward_train(ldm_module* m, const float* x_dev, const float* t_emb_dev, float* out_dev, int B, int H, int W,
                               float* bn_mean_dev, float* bn_var_dev, void* stream);
/* Kernel-level operator behind both: training-mode BatchNorm + the ResidualBlock's tail on an NHWC tensor of `dtype`
 * elements, h [B, HW, C] (C a multiple of 8 in bf16, of 4 in f32; 16-byte aligned):
 *   out = dropout(act(((h - mean) / sqrt(var + eps)) * gamma + beta + res) + tvec[b, :])
 * mean / var = per-channel batch statistics of h (biased variance; a two-stage sum in a fixed order, so the result
 * does not depend on scheduling).  res [B, HW, C] or NULL; tvec [B, C] f32 or NULL; out may be h, or NULL for the
 * statistics alone; act: 0 none, 1 SiLU, 2 ReLU; ws_dev: ldm_batchnorm_ws_floats(dtype, B * HW, C) floats;
 * mean_out_dev / var_out_dev [C] optional.  The dropout mask is the one described above with (H/4, W/4) = the
 * tensor's own pixels. */
long ldm_batchnorm_ws_floats(int dtype, long pixels, int channels);
int ldm_batchnorm_act(int dtype, const void* h_dev, const void* res_dev, void* out_dev, int B, long HW, int C,
                      const float* gamma_dev, const float* beta_dev, float eps, int act, const float* tvec_dev,
                      float dropout_p, unsigned long long seed, float* ws_dev, float* mean_out_dev, float* var_out_dev,
                      void* stream);

/* ---- the step after the loop: Vae::decode and the 8-bit image ------------------------------- */
/* SimpleDecoder::new(output, latent_dimen), src/main.rs:108-147.  The module takes its parameters through
 * ldm_module_set_param / ldm_module_finalize / ldm_module_destroy like the others: "transfer.weight"
 * [latent_dimen*16, 32768], "transfer.bias", "reverse1.weight" [128,64,4,4], "reverse1.bias", "bn_1.gamma" /
 * "beta" / "running_mean" / "running_var" [64], "reverse2.weight" [64,32,4,4], "reverse2.bias", "bn_2.*" [32],
 * "reverse3.weight" [32,output,3,3], "reverse3.bias".  latent_dimen a multiple of 8 up to 64. */
int ldm_vae_decoder_create(ldm_module** out, int dtype, int output_channels, int latent_dimen);
/* SimpleDecoder::forward (= Vae::decode, :208-210), src/main.rs:149-169: Linear + ReLU, reshape to
 * [B,128,16,16], ConvTranspose 4x4/s2 + BatchNorm + ReLU twice, ConvTranspose 3x3/s1, tanh.  Inference
 * semantics (running statistics, no dropout).  z [B,latent_dimen,4,4], img [B,output_channels,64,64]. */
int ldm_vae_decode(ldm_module* decoder, const float* z_dev, float* img_dev, int B, void* stream);
/* The pixel arithmetic of save_as_image, src/main.rs:1297-1308, for every image of the batch:
 * out[b,y,x,c] = (u8) clamp((img[b,c,y,x] + 1) * 127.5, 0, 255) (f32, truncating; NaN gives 0).
 * img [B,C,H,W] f32, out [B,H,W,C] u8.  (The reference writes image 0 to a PNG file; files are the caller's.) */
int ldm_to_rgb8(const float* img_dev, unsigned char* out_dev, int B, int C, int H, int W, void* stream);

/* ---- the step before the path in training: forward process and the loss value (no backward) -- */
/* The two scalars DiffusionModel::add_noise derives per timestep, src/main.rs:603-624 (alpha-bar made safe
 * and clamped to [1e-8, 0.9999]; sqrt clamped to [1e-8, 1]; sqrt of 1 - alpha clamped to [1e-8, 1]):
 * out_host [num_time][2] = {sqrt_alpha, sqrt_one_minus_alpha}. */
int ldm_add_noise_coefficients(int num_time, float* out_host);
/* DiffusionModel::add_noise, src/main.rs:584-630, with the noise supplied by the caller:
 * out[b] = x[b] * coef[t_b][0] + noise[b] * coef[t_b][1] (two products and a sum, f32).  x / noise / out
 * [B, per_sample] f32 (per_sample a multiple of 4), t_dev int[B] (outside [0, num_time): last row, :606),
 * coef_dev = the table above in device memory. */
int ldm_add_noise(const float* x_dev, const float* noise_dev, const int* t_dev, const float* coef_dev, int num_time,
                  float* out_dev, int B, long per_sample, void* stream);
/* The loss value of DiffusionModel::forward, src/main.rs:665-671: out_dev[0] = mean((a - b)^2) over n f32
 * elements, summed in a fixed order.  ws1024_dev: 1024 floats of device scratch. */
int ldm_mse(const float* a_dev, const float* b_dev, long n, float* ws1024_dev, float* out_dev, void* stream);
/* ldm_mse for callers that cannot allocate device memory (code under hipGraph capture, tight serving loops,
 * harnesses that forbid hipMalloc): no counterpart in the reference.  The 1024 partial sums live in a 4-KiB array
 * inside the library's code object (one copy per GPU, held until the process ends) instead of a workspace argument;
 * the call allocates nothing and returns the same value as ldm_mse, bit for bit.  n above LDM_MSE_NOALLOC_MAX_N is
 * rejected.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams) race.
 * The first call on a GPU looks the array's address up (hipGetSymbolAddress under a lock, which may load the code
 * object): a caller that captures its calls into a graph makes that first call, or calls ldm_mse_noalloc_prepare
 * with the GPU current, BEFORE the capture begins; after that the call only launches the two kernels. */
#define LDM_MSE_NOALLOC_MAX_N (1L << 36)
int ldm_mse_noalloc(const float* a_dev, const float* b_dev, long n, float* out_dev, void* stream);
int ldm_mse_noalloc_prepare(void);

/* ---- one convolution on NHWC tensors in the library's own layout (kernel-level operator) --- */
/* The burn layers behind it: Conv2d 3x3 stride 1 / stride 2 padding 1 (src/main.rs:348-366,423-452,
 * 486-492), ConvTranspose2d 4x4 stride 2 padding 1 (:460-480), Conv2d 1x1 on a channel concatenation
 * of two sources (:468-483 with the cat at :516,521).  kind: 0 conv3x3/s1, 1 conv3x3/s2,
 * 2 convT4x4/s2, 3 conv1x1.  weight_host: [cout,cin,3,3] | [cin,cout,4,4] | [cout,cin0+cin1]; bias [cout].
 * Tensors are NHWC device arrays of `dtype` elements (bf16 or f32): in0 [B,H,W,cin0], in1 [B,H,W,cin1]
 * or NULL, res (added before the activation) and out [B,Ho,Wo,cout].  Channel counts must be
 * multiples of 64.  act: 0 none, 1 SiLU, 2 ReLU. */
typedef struct ldm_conv ldm_conv;
int ldm_conv_create(ldm_conv** out, int kind, int dtype, int cin0, int cin1, int cout, const float* weight_host,
                    const float* bias_host);
int ldm_conv_forward(ldm_conv* c, const void* in0_dev, const void* in1_dev, const void* res_dev, void* out_dev,
                     int B, int Hin, int Win, int act, void* stream);
void ldm_conv_destroy(ldm_conv* c);
/* Diagnostics (no counterpart in the reference): the name of the convolution kernel that the calling thread's last
 * ldm_conv_forward (or any other convolution launch of the library) started, spelled as rocprofv3 lists it, e.g.
 * "void ldm::tapconv_kernel<unsigned short, 16, 64, 4, 2, 1, 3, 9, 2, 3, 9>(ldm::TapConvArgs)".  bench.py uses it to
 * pick the hardware-counter record that belongs to the kernel it timed. */
const char* ldm_last_kernel(void);

/* ---- one scheduler update on flat f32 arrays, src/main.rs:712-756,765 --------------------- */
/* coef5 = one row of ldm_ddpm_coefficients / ldm_ddim_coefficients.  DDPM: z <- (x0*c1 + z*c2) +
 * sigma*noise, or x0 when `last`; DDIM: z <- x0*c1 + eps*c2; x0 = (z - eps*s1m)/sa.  noise_dev may be
 * NULL for DDIM and for the last DDPM step. */
int ldm_sched_step(float* z_dev, const float* eps_dev, const float* noise_dev, long n, const float* coef5,
                   int mode, int last, void* stream);

/* ---- the sampling loop: DiffusionModel::denoising_process, src/main.rs:681-777 ------------ */
/* The loop runs on `batch` independent chains of shape [in_channels,H,W]; the trailing vae.decode
 * (:786) is outside this library.  ldm_sampler_create allocates the activation buffers, the state and the tables;
 * ldm_sampler_run allocates only when the caller injects per-step noise (noise_dev): a buffer for its NHWC copy,
 * on first use and when more steps are asked for (that synchronises the device).  A (mode, steps) pair seen for the
 * first time gets timestep and coefficient tables of its own (one blocking upload into fresh memory) and, with
 * use_graph, its own captured graph; both are kept, so alternating samplers or step counts costs nothing later.
 * At H = W = 4, the latent size of the reference's VAE, the sampler (and ldm_unet_forward) runs every
 * spatial layer as one matrix between whole images, built at ldm_module_finalize (LDM_DENSE4 above). */
int ldm_sampler_create(ldm_sampler** out, ldm_module* unet, int num_time, int batch, int H, int W);
/* mode LDM_DDPM visits i = num_time-1 .. 0; LDM_DDIM visits ddim_steps strided timesteps.
 * z0_dev   [batch,C,H,W] initial latent, or NULL: drawn from the counter-based generator.
 * noise_dev [steps,batch,C,H,W] per-step Gaussian noise (DDPM), or NULL: counter-based generator
 *          keyed by (seed, chain0 + chain index, step, element) -- independent of how chains are
 *          split over processes.
 * z_out_dev [batch,C,H,W].  z_trace_dev / eps_trace_dev: optional [steps,batch,C,H,W] records of
 *          each step's input latent and predicted noise (forces eager launches).
 * use_graph: replay the whole loop from one captured hipGraph (captured on first use). */
int ldm_sampler_run(ldm_sampler* s, int mode, int ddim_steps, const float* z0_dev, const float* noise_dev,
                    unsigned long long seed, unsigned int chain0, float* z_out_dev, float* z_trace_dev,
                    float* eps_trace_dev, int use_graph, void* stream);
void ldm_sampler_destroy(ldm_sampler* s);

/* The finiteness probes the reference runs outside its loop: on the final latent before vae.decode
 * (src/main.rs:780-784), on the decoded image (:788-792) and on the tensors of DiffusionModel::forward
 * (:647-651, 657-662, 672-676).  The reference reads the mean back and tests it; a mean is NaN or infinite as soon as
 * one element is, so this probe looks at the elements: *flag_dev = 1 when all n f32 elements of z_dev are finite,
 * 0 otherwise.  Two small launches on `stream`, no host synchronisation: the caller reads the flag when it wants
 * to and applies the reference's fallback (zeros, or the loss value 100).  z_dev may be any f32 view (4-byte aligned). */
int ldm_latent_is_finite(const float* z_dev, long n, int* flag_dev, void* stream);

/* Standard-normal field of the counter-based generator, [batch,C,H,W] f32 (C multiple of 4):
 * what ldm_sampler_run draws at `step` (0xFFFFFFFF = the initial latent). */
int ldm_philox_normal(float* out_dev, int batch, int C, int H, int W, unsigned long long seed,
                      unsigned int chain0, unsigned int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDM_HIP_H */
