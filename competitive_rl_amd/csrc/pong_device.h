// pong_device.h -- device-side types and the per-frame game update shared by the
// cPongDouble kernels (gfx950).  Compiled with -ffp-contract=off: the game's speeds
// are CPython floats (f64, one rounding per operation), so no FMA contraction.
//
// What each piece restates (reference paths relative to competitive_rl/):
//   serve_draw / ball serve ....... Ball.reset            pong/base_pong_env.py:314-320
//   decode_action / auto_action ... _step decode, auto_action        :116-134, :457-471
//   bat_move ...................... Bat.move                                  :412-418
//   frame_step .................... PongGame.step + Ball.move       :213-245, :325-361
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/crl.h"

namespace crl {

// SoA state of one shard of envs in HBM (DESIGN.md "HBM layout").  Passed by value.
struct PongSoA {
    double *speed_x, *speed_y;
    int32_t *ball_x, *ball_y, *bat_l, *bat_r, *score_l, *score_r, *rounds, *steps, *wrap_steps;
    uint32_t *serve_ctr;
    uint64_t *keep;         // [2][n]  MaxAndSkipEnv buffers as packed frames
    uint64_t *ring;         // [8][n]  frame-stack ring: plane p (0 oldest..3 newest), slot s -> ring[(2p+s)*n + i]
    uint64_t *obs_frames;   // [2][n]  frames the raster kernel draws this step (slot 0 only in raw mode)
    uint64_t *term_frames;  // [2][n]  frames of the last terminal observation
    float *real_reward;     // [n][2]
    int32_t *num_steps;     // [n]
    int32_t *bad_action;    // host-mapped flag: first action outside {0, 1, 2, 999} seen, stored as action + 1 (0 = none)
};

struct ServeSrc {
    uint64_t seed;
    int64_t env_id_base;
    const double *ru;  // replay stream or nullptr
    const uint8_t *rbx, *rby;
    int64_t per_env;
};

// The registers one env's game lives in while a kernel steps it.
struct PongEnv {
    double sx, sy;
    int32_t x, y, bl, br, score_l, score_r, rounds, steps;
    uint32_t serve_ctr;
};

static constexpr uint64_t kBlankFrame = 0xFFFF000000000000ull;  // score_l = score_r = 255

// crl_pong_frame packed little-endian into a u64: [ball_x:16][ball_y:16][bat_l:8][bat_r:8][sl:8][sr:8]
__host__ __device__ inline uint64_t pack_frame(int x, int y, int bl, int br, int sl, int sr) {
    return (uint64_t)(uint16_t)(int16_t)x | ((uint64_t)(uint16_t)(int16_t)y << 16) | ((uint64_t)(uint8_t)bl << 32) |
           ((uint64_t)(uint8_t)br << 40) | ((uint64_t)(uint8_t)sl << 48) | ((uint64_t)(uint8_t)sr << 56);
}

struct Frame {
    int x, y, bl, br, sl, sr;
};

__host__ __device__ inline Frame unpack_frame(uint64_t f) {
    Frame r;
    r.x = (int16_t)(f & 0xFFFF), r.y = (int16_t)((f >> 16) & 0xFFFF);
    r.bl = (int)((f >> 32) & 0xFF), r.br = (int)((f >> 40) & 0xFF);
    r.sl = (int)((f >> 48) & 0xFF), r.sr = (int)((f >> 56) & 0xFF);
    return r;
}

__device__ inline uint64_t frame_of(const PongEnv &e) {
    return pack_frame(e.x, e.y, e.bl, e.br, e.score_l, e.score_r);
}

// Philox4x32-10 keyed by the seed, counter = (global env id, serve number).
__device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}

__device__ inline void serve(PongEnv &e, const ServeSrc &s, int64_t env) {
    double u;
    int bx, by;
    if (s.per_env > 0) {
        int64_t j = env * s.per_env + (int64_t)((uint64_t)e.serve_ctr % (uint64_t)s.per_env);
        u = s.ru[j], bx = s.rbx[j], by = s.rby[j];
    } else {
        uint64_t gid = (uint64_t)(s.env_id_base + env);
        uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), e.serve_ctr, 0x504F4E47u};
        philox4x32_10(c, (uint32_t)s.seed, (uint32_t)(s.seed >> 32));
        uint64_t m = (((uint64_t)c[0] << 32) | c[1]) >> 11;
        u = (double)m * (1.0 / 9007199254740992.0);
        bx = c[2] & 1, by = c[3] & 1;
    }
    e.serve_ctr++;
    e.x = 78, e.y = 112;
    const double a = 4 * 0.3, b = 4.0;
    double sp = a + (b - a) * u;  // random.uniform(a, b)
    e.sx = bx ? 4.0 : -4.0;
    e.sy = by ? sp : -sp;
}

__device__ inline void round_reset(PongEnv &e, const ServeSrc &s, int64_t env) {
    serve(e, s, env);
    e.rounds++;
    e.steps = 0;
    e.bl = 107, e.br = 107;
}

__device__ inline void game_reset(PongEnv &e, const ServeSrc &s, int64_t env) {
    e.score_l = e.score_r = 0;
    round_reset(e, s, env);
    e.rounds = 0;
}

__device__ inline int auto_action(double ball_sx, int bat_cy, int ball_cy) {
    const int arena_cy = CRL_PONG_TOP + (CRL_PONG_W >> 1);
    int d = 0;
    if (ball_sx < 0) d = bat_cy < arena_cy ? 1 : (bat_cy > arena_cy ? -1 : 0);
    else if (ball_sx > 0) d = bat_cy < ball_cy ? 1 : -1;
    return d;
}

__device__ inline int decode_action(int a, double sx_for_side, int bat_y, int ball_y) {
    if (a == CRL_PONG_CHEAT) return auto_action(sx_for_side, bat_y + (CRL_PONG_BAT_H >> 1), ball_y + (CRL_PONG_BALL >> 1));
    // BAT_DIRECTIONS[a] (base_pong_env.py:13); anything else fails the reference's `action_space.contains` assert
    // (:42) -- here the bat stays put and the step kernel raises the context's flag (CRL_EACTION)
    return (unsigned)a < 3u ? a - 1 : 0;
}
__device__ inline bool action_ok(int a) { return (unsigned)a < 3u || a == CRL_PONG_CHEAT; }

__device__ inline int bat_move(int32_t &y, int dir) {
    int mv = dir * 4;
    y += mv;
    if (y + CRL_PONG_BAT_H > CRL_PONG_BOTTOM) y = CRL_PONG_BOTTOM - CRL_PONG_BAT_H;
    else if (y < CRL_PONG_TOP) y = CRL_PONG_TOP;
    return mv;
}

// One raw frame.  Returns done; r_l / r_r are the two players' rewards.
__device__ inline bool frame_step(PongEnv &e, int a_l, int a_r, const ServeSrc &s, int64_t env, int &r_l, int &r_r) {
    const int dr = decode_action(a_r, e.sx, e.br, e.y);
    const int dl = decode_action(a_l, -e.sx, e.bl, e.y);
    e.steps += 1;
    const int mv_l = bat_move(e.bl, dl);
    const int mv_r = bat_move(e.br, dr);

    const int batL_right = CRL_PONG_BATL_X + CRL_PONG_BAT_W, batR_left = CRL_PONG_BATR_X;
    const int prev_left = e.x, prev_right = e.x + CRL_PONG_BALL;
    double sx = e.sx, sy = e.sy;
    const double y_on_r = (double)(batR_left - prev_right) / sx * sy + (double)e.y;
    const double y_on_l = (double)(batL_right - prev_left) / sx * sy + (double)e.y;
    int x = (int)((double)e.x + sx);  // Rect stores truncate toward zero
    int y = (int)((double)e.y + sy);
    if (sy < 0 && y <= CRL_PONG_TOP) {
        sy = -sy, y = CRL_PONG_TOP;
    } else if (sy > 0 && y + CRL_PONG_BALL >= CRL_PONG_BOTTOM) {
        sy = -sy, y = CRL_PONG_BOTTOM - CRL_PONG_BALL;
    } else if (sx < 0 && x <= batL_right && y_on_l + 4 >= (double)e.bl && y_on_l <= (double)(e.bl + CRL_PONG_BAT_H) &&
               prev_left > batL_right) {
        sx = -sx, sy = sy + (double)mv_l * 0.7;
        x = batL_right, y = (int)y_on_l;
    } else if (sx > 0 && x + CRL_PONG_BALL >= batR_left && y_on_r + 4 >= (double)e.br &&
               y_on_r <= (double)(e.br + CRL_PONG_BAT_H) && prev_right < batR_left) {
        sx = -sx, sy = sy + (double)mv_r * 0.7;
        x = batR_left - CRL_PONG_BALL, y = (int)y_on_r;
    }
    e.x = x, e.y = y, e.sx = sx, e.sy = sy;

    r_l = r_r = 0;
    if (x < 0) {
        e.score_r++, r_l = -1, r_r = 1;
        round_reset(e, s, env);
    } else if (x + CRL_PONG_BALL > CRL_PONG_W) {
        e.score_l++, r_l = 1, r_r = -1;
        round_reset(e, s, env);
    } else if (e.steps > CRL_PONG_MAX_STEPS) {
        round_reset(e, s, env);
    }
    return e.rounds >= CRL_PONG_MAX_ROUNDS;
}

__device__ inline PongEnv load_env(const PongSoA &s, int64_t i) {
    PongEnv e;
    e.sx = s.speed_x[i], e.sy = s.speed_y[i];
    e.x = s.ball_x[i], e.y = s.ball_y[i], e.bl = s.bat_l[i], e.br = s.bat_r[i];
    e.score_l = s.score_l[i], e.score_r = s.score_r[i], e.rounds = s.rounds[i], e.steps = s.steps[i];
    e.serve_ctr = s.serve_ctr[i];
    return e;
}

__device__ inline void store_env(const PongSoA &s, int64_t i, const PongEnv &e) {
    s.speed_x[i] = e.sx, s.speed_y[i] = e.sy;
    s.ball_x[i] = e.x, s.ball_y[i] = e.y, s.bat_l[i] = e.bl, s.bat_r[i] = e.br;
    s.score_l[i] = e.score_l, s.score_r[i] = e.score_r, s.rounds[i] = e.rounds, s.steps[i] = e.steps;
    s.serve_ctr[i] = e.serve_ctr;
}

// launchers (defined in the .hip files, called from crl_api.hip)
struct PongMode {
    bool wrapped, single, replicate;  // MaxAndSkip path; cPong-v0 (AutoBat on the right); FrameStack fill
};
void launch_pong_reset(const PongSoA &s, const ServeSrc &src, int64_t n, PongMode mode, hipStream_t st);
void launch_pong_dynamics(const PongSoA &s, const ServeSrc &src, const int32_t *actions, int64_t n, PongMode mode,
                          float *rew, uint8_t *done, hipStream_t st);
// out[(6, 7) * count + k] = frames[(0, 1) * n + idx[k]]: the pairs of `count` device-resident env indices as the newest
// plane of a single-plane ring (planes 0..2 are not written: K = 1 never reads them)
void launch_pong_gather_frames(const uint64_t *frames, const int64_t *idx_dev, int64_t count, int64_t n, uint64_t *ring_out,
                               hipStream_t st);
void launch_pong_raster_raw(const uint64_t *frames, int64_t n, const uint8_t *atlas_rgb, int ink_row0, int ink_row1,
                            uint8_t *obs, int views, hipStream_t st);

// Byte offsets of the dense INTER_AREA tables inside the blob the gray kernel stages into
// LDS: xa/ya = float[5][R] weights (tap k of output index d at [k*R + d]); xs0/xn, ys0/yn =
// u8[R] first source index and tap count; xf/xl = u8[160], yf/yl = u8[210] first/last
// output index fed by a source col/row.  fast_ok: every tap of an output row that a court
// rectangle can touch lies in a source row without score ink.
struct GrayTabOfs {
    int xa, ya, xs0, xn, ys0, yn, xf, xl, yf, yl, total, fast_ok, max_taps;
    int xa255, rowstatic, colstatic;  // 255*xalpha (f32 [5][R]); static bits of row_pack / col_pack (u32 [R])
    int box32;  // byte offset (past `total`, global memory only) of int32 copies of xf|xl|yf|yl for scalar loads
};
// FrameStackTensor fused into the draw (crl_step_stack / crl_draw_stack; reference utils/utils.py:145-173 called from step_envs :23-60): the k
// planes of one agent's rolling stack drawn from the ring by the launch that draws the observation, instead of a separate roll-and-append pass.
struct GrayStack {
    uint8_t *out;   // (n, k, R, R) of agent `view`; nullptr = no stack
    int k, view;    // planes oldest to newest = ring planes 4 - k .. 3
    int f32;        // element type: 0 u8, 1 float32 (the context's float values)
    int valid;      // updates since the stack's reset(), capped at k: the k - valid oldest planes are zeros
    int alias;      // 1: the observation tensor's (view, newest plane) tile is not written -- its reader takes the stack's newest plane
};
struct GrayParams {
    const uint64_t *ring;    // [8][n] frame pairs of the 4 stack planes; plane 3 = newest (K=1 draws only it)
    int64_t n;
    int R, K, views;
    const uint8_t *atlas_gray;  // [22*22][34][160]
    const uint8_t *band;        // pre-resized top band [22*22][2 views][band_rows][R]
    int band_rows;
    const int32_t *xofs, *yofs;     // [R+1] first tap of each output col / row
    const int32_t *xsi, *ysi;       // source index per tap
    const float *xalpha, *yalpha;   // weight per tap
    uint8_t *obs;                   // [n][2][K][R][R]
    int obs_f32;                    // crl_obs_dtype of obs: 1 = float32, same values, widened in the store epilogue; 2 = the reference's unrounded float32 path
    void *hdr;                      // scratch for the address-linear writer: 64 B per (env, view, plane) tile, or nullptr
    // CRL_OBS_F32_REF: the planes of a court WITHOUT ball and bats, per score pair (pong_raster_gray.hip pong_gray_f32ref_kernel)
    const float *f32_top;           // [484][2 views][2: unrounded, rounded][band_rows][R] output rows fed by the score band
    const float *f32_bot;           // [2][R - f32_bot0][R] output rows fed by the white band under the court (score-independent)
    int f32_bot0;                   // first output row with a tap under the court
    int f32_xtaps, f32_ytaps;       // entries of xsi / ysi (the kernel stages the tap tables in LDS)
    int f32_map_row0, f32_map_rows; // output rows fed by the court's source rows (where a ball or a bat can change a pixel)
    GrayStack stack;                // fused FrameStackTensor (out == nullptr: none); obs may then be nullptr (stack only)
};
void launch_pong_gray_f32ref_tables(const GrayParams &p, float *top, float *bot, hipStream_t st);
void launch_pong_raster_gray(const GrayParams &p, hipStream_t st);

}  // namespace crl
