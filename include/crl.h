/*
 * crl.h -- C ABI of the MI355X-native vector-env backend (libcrl_hip.so).
 *
 * The reference (ucla-rlcourse/competitive-rl) is pure Python and has NO FFI:
 * its boundary for this path is the Python VecEnv protocol.  Each entry point
 * below names the reference interface it stands in for (file:line relative to
 * /root/reference/competitive_rl/).  The Python host mirror of that protocol
 * (competitive_rl_amd/vec_env.py) is the only in-tree caller; INTEGRATION.md
 * shows the ctypes stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative CRL_E* code (crl_last_error() has the text); all
 * device work is ordered on the caller's HIP stream (`stream` is a
 * hipStream_t passed as void*: whatever the caller enqueues there afterwards
 * sees the results; a CarRacing step forks to the context's own streams and
 * joins them back before it returns); a context is bound to one GPU and is not
 * thread-safe; `*_dev` pointers are device memory owned by the caller
 * (e.g. torch tensors), `*_host` pointers are host memory.
 */
#ifndef CRL_H_
#define CRL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry of cPongDouble-v0 (pong/base_pong_env.py:158-211) */
#define CRL_PONG_W 160
#define CRL_PONG_H 210
#define CRL_PONG_TOP 34          /* arena top == top_border_thickness          */
#define CRL_PONG_BOTTOM 194      /* arena bottom = 34 + window_width (sic)     */
#define CRL_PONG_BALL 4
#define CRL_PONG_BAT_W 5
#define CRL_PONG_BAT_H 15
#define CRL_PONG_BATL_X 16
#define CRL_PONG_BATR_X 139
#define CRL_PONG_MIRROR_ROW 25   /* base_pong_env.py:153-154: rows >= 25 flip  */
#define CRL_PONG_MAX_ROUNDS 21   /* pong/register.py:20-22                     */
#define CRL_PONG_MAX_STEPS 10000 /* base_pong_env.py:171                       */
#define CRL_PONG_CHEAT 999       /* base_pong_env.py:9                         */
#define CRL_PONG_FRAME_BYTES (CRL_PONG_H * CRL_PONG_W * 3) /* 100 800 */
/* score band: 22 x 22 (score_l, score_r) images of rows 0..33, gray u8        */
#define CRL_PONG_ATLAS_SCORES 22
#define CRL_PONG_ATLAS_BYTES (22 * 22 * CRL_PONG_TOP * CRL_PONG_W)

enum { CRL_OK = 0, CRL_EINVAL = -1, CRL_EHIP = -2, CRL_ENOMEM = -3, CRL_ESTATE = -4,
       /* an earlier crl_step was given a Pong action outside {0, 1, 2, 999}: the reference asserts
          action_space.contains(action) (pong/base_pong_env.py:42).  Device-resident actions are
          checked by the step kernel; the flag reaches the host without a sync, so the error is
          reported ONCE, by the first crl_step / crl_reset / crl_check that sees it (that call does no
          work; the offending bat did not move), and cleared: the next call proceeds. */
       CRL_EACTION = -5 };

enum crl_env_kind {
    CRL_ENV_PONG_DOUBLE = 1, /* cPongDouble-v0                                                    */
    CRL_ENV_CAR_DOUBLE = 2,  /* cCarRacingDouble-v0                                               */
    /* cPong-v0 (PongSinglePlayerEnv, pong/base_pong_env.py:16-82): the right bat is the AutoBat
       (:445-454) = action 999 on that side; actions int32 (N), one view, obs (N,1,...) raw or
       (N,K,R,R) wrapped, rewards f32 (N) = the left player's */
    CRL_ENV_PONG_SINGLE = 3,
    /* cCarRacing-v0: CarRacing(num_player=1) (car_racing/register.py:11-17, 29-40): one car,
       actions f32 (N,1,2), obs (N,K,96,96), rewards f32 (N,1) */
    CRL_ENV_CAR_SINGLE = 4,
};
/* crl_opts.flags */
#define CRL_FLAG_CAR_NO_CONTACTS 2  /* cCarRacingDouble: skip the car-car contact constraints (cars pass
                                      through each other); default is to solve them */
#define CRL_FLAG_CAR_FMA 4          /* cCarRacing*: b2World.Step's island solver (car_racing_multi_players.py:600 -> b2Island::Solve:
                                      integrators, the 180 velocity and <= 60 position iterations of joints and contacts)
                                      evaluates every a*b+c in ONE fused multiply-add instead of Box2D's two roundings:
                                      0.46 x the instructions of the step's longest dependent chain.  Checked at tolerance 0
                                      against the CPU checker's -DCRL_FMA build (the same sites as MAD / NMAD); one step away from the
                                      checker's libm build by <= 2.7e-5 relative on velocities (default: 5.6e-6),
                                      which is why it is opt-in */
#define CRL_FLAG_STACK_REPLICATE 1 /* FrameStack wrapper semantics (utils/atari_wrappers.py:243-247):
                                      reset fills all K planes with the first frame, instead of
                                      FrameStackTensor's zeroed history */

/* ---- cCarRacingDouble-v0 (car_racing/car_racing_multi_players.py:54-88) */
#define CRL_CAR_MAX_TILES 512 /* tiles of one track (reference tracks: 230-380) */
#define CRL_CAR_OBS 96        /* STATE_W = STATE_H = 96 */
/* The pre-rastered observation map (render_road_for_observation_map, car_racing_multi_players.py:732-755):
   the reference draws into a 10000 x 10000 surface; everything that is not grass lies inside the window
   [CRL_CAR_MAP_ORG, CRL_CAR_MAP_ORG + CRL_CAR_MAP_W)^2 of it (1.7636 px per world unit, origin at 5000),
   which is what a context keeps per env: 7-colour palette, 4 bits per pixel, 739 328 bytes. */
#define CRL_CAR_MAP_ORG 4392
#define CRL_CAR_MAP_W 1216
/* reward read-out of the indicator strip: bitmaps of "%05.0f" % r for r = -999..2000 plus "-0000",
   10 rows of 32 bits each (competitive_rl_amd/assets/car_reward_text.npz) */
#define CRL_CAR_TEXT_STRINGS 3001
#define CRL_CAR_TEXT_ROWS 10
#define CRL_CAR_TEXT_RMIN (-999)

enum crl_obs_mode {
    /* raw env: obs (N,2,210,160,3) u8, 1 step = 1 frame
       (PongDoublePlayerEnv._step, pong/base_pong_env.py:113-142) */
    CRL_OBS_RAW_RGB = 0,
    /* make_env_a2c_atari stack: skip-4 + max-2 + gray + INTER_AREA resize
       (utils/atari_wrappers.py:40-53,89-219), obs (N,2,K,R,R) u8 with
       FrameStackTensor roll/zero-on-done semantics for K>1 (utils/utils.py:145-173) */
    CRL_OBS_GRAY_RESIZED = 1,
};

/* One raster-ready frame of the game, 8 bytes (what Arena/Ball/Bat/Scoreboard.draw
 * read: base_pong_env.py:259-266).  ball_x/ball_y are int16 because a bat hit can
 * place the ball a few px outside the arena rows (it is clipped when drawn). */
typedef struct crl_pong_frame {
    int16_t ball_x, ball_y;
    uint8_t bat_l_y, bat_r_y;
    uint8_t score_l, score_r; /* score_l == 255 marks a BLANK (all-zero) plane */
} crl_pong_frame;

/* Per-env state exchanged by crl_get_state / crl_set_state (host side, AoS).
 * On the device the same fields live in SoA arrays (DESIGN.md "HBM layout"). */
typedef struct crl_pong_env_state {
    double speed_x, speed_y;       /* Ball._speed_x/_speed_y (f64)                  */
    int32_t ball_x, ball_y;        /* Ball._rect.x/.y                               */
    int32_t bat_l_y, bat_r_y;      /* Bat._rect.y                                   */
    int32_t score_l, score_r;      /* PongGame._score_left/_right                   */
    int32_t num_rounds, num_steps; /* PongGame._num_rounds/_num_steps               */
    uint32_t serve_ctr;            /* serves drawn so far (RNG / replay cursor)     */
    int32_t wrap_steps;            /* ClipRewardEnv._steps (atari_wrappers.py:169)  */
    crl_pong_frame keep[2];        /* MaxAndSkipEnv._obs_buffer as frames (:104-116)*/
    crl_pong_frame hist[3][2];     /* frame_stack>1: the 3 older planes of the stack,
                                      oldest first, each the (keep0, keep1) pair that
                                      produced it; BLANK = plane zeroed by a done
                                      (FrameStackTensor.update, utils/utils.py:158-170) */
} crl_pong_env_state;              /* 120 bytes */

typedef struct crl_opts {
    int32_t env_kind;    /* crl_env_kind                                             */
    int32_t obs_mode;    /* crl_obs_mode                                             */
    int32_t resized_dim; /* Pong: R = 84 or 42 (make_envs.py:67 resized_dim), 0 for raw;
                            CarRacing: ignored                                       */
    int32_t frame_stack; /* K planes per agent in GRAY_RESIZED mode (1 or 4)         */
    int64_t num_envs;    /* envs owned by THIS context (one shard)                   */
    int64_t env_id_base; /* global id of env 0 of this shard: RNG is keyed by global
                            id so results do not depend on how envs are sharded      */
    uint64_t seed;       /* make_envs(seed=...)                                      */
    int32_t device;      /* HIP device ordinal                                       */
    int32_t flags;       /* CRL_FLAG_*                                                  */
    int32_t action_repeat; /* CarRacing(action_repeat=...) (car_racing_multi_players.py:162,576):
                              0 or 1 = none, at most 16; Pong: must be 0                */
    int32_t done_policy;   /* CarRacing: crl_car_done_policy; Pong: must be 0           */
    int32_t obs_dtype;     /* crl_obs_dtype of obs_dev (GRAY_RESIZED Pong contexts; others: U8) */
    int32_t reserved;      /* must be 0                                                 */
} crl_opts;

/* Which cars end an env's episode under the VecEnv (CarRacing contexts) */
enum crl_car_done_policy {
    /* make_car_racing_double: FlattenMultiAgentObservation returns any(done.values())
       (utils/atari_wrappers.py:329-330) */
    CRL_CAR_DONE_ANY = 0,
    /* make_competitive_car_racing: CarRacingWrapper returns d[0]
       (car_racing/make_competitive_car_racing.py:24-33); a finished car 1 stays in the world,
       frozen (car_racing_multi_players.py:578-579), while car 0 drives on */
    CRL_CAR_DONE_CAR0 = 1,
};
/* Element type of the observation tensor handed to crl_reset / crl_step / crl_render */
enum crl_obs_dtype {
    CRL_OBS_U8 = 0,
    /* DummyVecEnv's observation buffers are float32 holding 0..255 (utils/dummy_vec_env.py:37-44;
       SURVEY 8d config-3 variant, 225 792 B/env): same values, widened in the raster's store */
    CRL_OBS_F32 = 1,
    /* The reference's own float32 values.  Old gym's Box defaults to float32, so MaxAndSkipEnv's buffers are float32
       (utils/atari_wrappers.py:104-116) and WarpFrame.parse_single_frame (:215-219) runs cv2.cvtColor / cv2.resize on FLOAT
       frames during step(): the observation is the UNROUNDED INTER_AREA average of gray = R*0.299f + G*0.587f + B*0.114f
       (e.g. 254.99998 where the uint8 path says 255); reset() -- and the auto-reset of a finished env -- goes through the
       uint8 image and yields rounded values.  A plane whose two kept frames are the same frame is such a reset observation.
       A per-score-pair table of the court without ball and bats + an exact re-draw of the ~150 pixels around them, patched into the pieces
       before they are stored: 2.8 ms per step at 65 536 envs x (2, 4, 84, 84) (23.5 M env-steps/s), HBM traffic 1.005 x the tensor;
       CRL_OBS_F32 (the uint8 values, widened) takes 2.6 ms. */
    CRL_OBS_F32_REF = 2,
};

typedef struct crl_ctx crl_ctx;

/* make_envs(...) -> VecEnv construction (make_envs.py:67-118; DummyVecEnv.__init__
 * dummy_vec_env.py:26-46).  `score_atlas_host`: CRL_PONG_ATLAS_BYTES gray values of
 * the top band for every score pair (Scoreboard.draw, base_pong_env.py:474-487).  CarRacing
 * contexts take the reward-text bitmaps instead (uint32 [CRL_CAR_TEXT_STRINGS][CRL_CAR_TEXT_ROWS],
 * draw_text, car_racing/pygame_rendering.py:16-18) or NULL to leave the text out. */
int crl_create(const crl_opts *opts, const uint8_t *score_atlas_host, crl_ctx **out);

/* VecEnv.close (dummy_vec_env.py:77-79; idempotent like subproc_vec_env.py:131-141) */
void crl_destroy(crl_ctx *ctx);

/* VecEnv.seed(seed): env i gets seed + i (dummy_vec_env.py:65-69).  In the reference
 * this never reaches Pong's RNG (_seed is a no-op, base_pong_env.py:38-39); here it
 * re-keys the counter-based serve sampler. */
int crl_seed(crl_ctx *ctx, uint64_t seed);

/* VecEnv.reset() (dummy_vec_env.py:71-75): resets every env, writes the first
 * observation into obs_dev (layout per obs_mode). */
int crl_reset(crl_ctx *ctx, uint8_t *obs_dev, void *stream);

/* VecEnv.step(actions) = step_async + step_wait (base_vec_env.py:178-187,
 * dummy_vec_env.py:48-63) with auto-reset.  actions_dev: Pong int32 (N,2), each 0/1/2 or
 * 999; CarRacing float32 (N,2,2) = per car (steer, gas-or-brake) in [-1,1]; CarRacing obs is
 * (N,2,96,96) u8, rew (N,2) per-car step rewards, done (N) = any car done or TimeLimit.  obs_dev: per obs_mode.  rew_dev: f32 (N,2) (raw: game reward; wrapped:
 * np.sign of the 4-frame sum).  done_dev: u8 (N).  Any output pointer may be NULL
 * to skip that output (obs_dev NULL = dynamics only). */
int crl_step(crl_ctx *ctx, const void *actions_dev, uint8_t *obs_dev, float *rew_dev,
             uint8_t *done_dev, void *stream);

/* ---- step_envs' observation stack fused into the step (utils/utils.py:23-60 calls FrameStackTensor.update :158-170 on the
 * learner's observation with mask = 1 - done right after envs.step).  A GRAY_RESIZED Pong context keeps the descriptors of every
 * env's last four planes with exactly that history rule (planes of an episode that ended are erased, the first observation of the
 * next one is the newest plane), so the stack the trainer would roll and append -- 13.4 GB of traffic per step for 65 536 x
 * (4, 84, 84) float32 -- is DRAWN by the launch that draws the observation: written once, never read.
 *   stack_dev     (N, planes, R, R) of agent `agent`, planes oldest to newest, u8 or f32 per `dtype` (CRL_OBS_U8 | CRL_OBS_F32; a
 *                 float32 context -- CRL_OBS_F32 / CRL_OBS_F32_REF -- writes its own float values and needs dtype = CRL_OBS_F32)
 *   valid_planes  updates since the trainer's FrameStackTensor.reset(), capped at `planes`: the older planes are zeros
 *   alias_newest  1: the (agent, newest plane) tile of obs_dev is NOT written -- the caller hands out the stack's newest plane as
 *                 that agent's observation (needs a context with frame_stack = 1 and equal element types)
 * Contexts with CRL_FLAG_STACK_REPLICATE (the FrameStack wrapper's history) and raw contexts refuse (CRL_ESTATE). */
typedef struct crl_stack_desc {
    void *stack_dev;
    int32_t planes;       /* k = 1..4 */
    int32_t dtype;        /* crl_obs_dtype: CRL_OBS_U8 or CRL_OBS_F32 */
    int32_t agent;        /* 0 = the learner (step_envs takes obs[0], utils/utils.py:55-58) */
    int32_t valid_planes;
    int32_t alias_newest;
    int32_t reserved;     /* must be 0 */
} crl_stack_desc;
/* crl_step + the stack: envs.step(actions) followed by frame_stack_tensor.update(obs[0], 1 - done) (utils/utils.py:30,57-58).
 * stack == NULL is crl_step.  obs_dev may be NULL (stack only). */
int crl_step_stack(crl_ctx *ctx, const void *actions_dev, uint8_t *obs_dev, float *rew_dev, uint8_t *done_dev,
                   const crl_stack_desc *stack, void *stream);
/* The stack of the CURRENT state without stepping: frame_stack_tensor.update(envs.reset()) of the training scripts, and what the
 * host mirror compares a trainer's own tensor with before it binds it.  obs_dev (optional) is re-drawn as by crl_render. */
int crl_draw_stack(crl_ctx *ctx, uint8_t *obs_dev, const crl_stack_desc *stack, void *stream);

/* A hipEvent_t of the caller's (NULL: none) that every crl_step / crl_step_stack records on the step's stream right BEHIND the kernel
 * that writes rew_dev / done_dev and IN FRONT of the observation's draw.  The reference's step_envs walks the done flags on the host
 * after every step (utils/utils.py:33-42); a stream that waits for this event can copy them out while the step's 1-2 ms of raster
 * still run, so the host's books and the next step's launch overlap the draw instead of following it.  Pong contexts.  The event is
 * the caller's: it must belong to the context's device and stay alive until the context is destroyed or the event is unset (NULL). */
int crl_set_flags_event(crl_ctx *ctx, void *event);

/* info[i]["real_reward"], info[i]["num_steps"] (ClipRewardEnv.step,
 * atari_wrappers.py:175-181) as device arrays valid until the next step:
 * real_reward f32 (N,2), num_steps i32 (N). */
int crl_info(crl_ctx *ctx, const float **real_reward_dev, const int32_t **num_steps_dev);
/* Same data copied (device to device, on `stream`) into caller-owned arrays; either
 * destination may be NULL. */
int crl_copy_info(crl_ctx *ctx, float *real_reward_out_dev, int32_t *num_steps_out_dev, void *stream);

/* info[i]["terminal_observation"] (dummy_vec_env.py:55-57), produced lazily: renders,
 * for `count` env indices (host array), the observation the episode ended on at the
 * most recent step where that env was done.  out_dev: raw (count,2,210,160,3) or
 * wrapped (count,2,R,R) u8; CarRacing (count,players,96,96) u8 = the frames drawn just before the
 * auto-reset (with a K-stack the caller prepends the K-1 newest planes it already holds). */
int crl_terminal_observation(crl_ctx *ctx, const int64_t *env_idx_host, int64_t count,
                             uint8_t *out_dev, void *stream);

/* The same with the env indices in DEVICE memory (e.g. torch.nonzero(done)): descriptors are gathered
 * by a kernel and drawn on `stream` -- no device-to-host copy, no allocation (a scratch buffer grows
 * on demand), no synchronisation.  This is what the host mirror's lazy infos use to fetch every
 * finished env of a step in one call. */
int crl_terminal_observation_dev(crl_ctx *ctx, const int64_t *env_idx_dev, int64_t count,
                                 uint8_t *out_dev, void *stream);

/* Synchronises `stream` and reports (then clears) a pending CRL_EACTION. */
int crl_check(crl_ctx *ctx, void *stream);

/* Parity tests + checkpoint/resume: whole-state copy, host AoS <-> device SoA.
 * Synchronises `stream`. */
int crl_get_state(crl_ctx *ctx, crl_pong_env_state *state_host, int64_t first, int64_t count,
                  void *stream);
int crl_set_state(crl_ctx *ctx, const crl_pong_env_state *state_host, int64_t first,
                  int64_t count, void *stream);

/* Replay mode for the serve sampler: per env a recorded stream of
 * `per_env` draws (u in [0,1), bit_x, bit_y); serve k of env i uses entry
 * [i*per_env + (k % per_env)].  Pass per_env = 0 to return to the Philox sampler. */
int crl_set_replay(crl_ctx *ctx, const double *u_host, const uint8_t *bx_host,
                   const uint8_t *by_host, int64_t per_env);

/* Render arbitrary frames (tests, get_images()/render(), base_vec_env.py:189-217).
 * frames_host: `count` frames; out_dev: (count,2,210,160,3) u8. */
int crl_render_raw(crl_ctx *ctx, const crl_pong_frame *frames_host, int64_t count,
                   uint8_t *out_dev, void *stream);

/* ---- observations by descriptor (BASELINE config #5: every GPU ends up with all N observations)
 * The reference's "gather" is _flatten_obs over the workers' pixel arrays (utils/subproc_vec_env.py:188-222).  Here a frame
 * is a pure function of its 8-byte descriptors, so shards exchange those: crl_obs_descriptors copies the descriptors the
 * current observation was drawn from -- crl_pong_frame [8][N]: stack plane p (0 oldest .. 3 newest), kept frame s (the two
 * frames MaxAndSkipEnv takes the maximum of) in row 2p + s; raw contexts and frame_stack = 1 use rows 6 and 7 -- and
 * crl_render_frames_dev draws `count` observations from descriptors in that layout ([8][count], anyone's), out_dev as
 * crl_step's obs_dev for `count` envs.  64 bytes per env instead of 56 448 (fused 4-stack) or 201 600 (raw). */
int crl_obs_descriptors(crl_ctx *ctx, crl_pong_frame *desc_out_dev, void *stream);
int crl_render_frames_dev(crl_ctx *ctx, const crl_pong_frame *desc_dev, int64_t count, uint8_t *out_dev, void *stream);

/* Re-draws the CURRENT state into obs_dev without stepping (after crl_set_state /
 * crl_car_set_state, or for VecEnv.render): same layout as crl_step's obs_dev. */
int crl_render(crl_ctx *ctx, uint8_t *obs_dev, void *stream);

/* Bytes of one env's observation in the context's obs_mode. */
int64_t crl_obs_bytes_per_env(const crl_ctx *ctx);

/* Launch-level timing hook for bench.py: wraps the `which`-th kernel of crl_step
 * (0 = dynamics, 1 = raster) in hipEvents on the launch stream and accumulates.
 * crl_kernel_time_ms() synchronises and returns total ms and launch count. */
int crl_kernel_timing(crl_ctx *ctx, int enable);
int crl_kernel_time_ms(crl_ctx *ctx, int which, double *total_ms, int64_t *launches);
/* The same with the longest launch, and one more slot for CarRacing contexts: which = 2 is the touching solve (car_touch_kernel, the
 * island solve of the envs whose cars touch: b2World.Step, car_racing_multi_players.py:600), the kernel a steady-state step ends with. */
int crl_kernel_time_stats(crl_ctx *ctx, int which, double *total_ms, int64_t *launches, double *max_ms);

/* ---- cCarRacingDouble state exchange (parity tests, checkpoint) ---------------------- */
typedef struct crl_car_body { /* b2Body: centre of mass, angle, velocities */
    float cx, cy, a, vx, vy, w;
} crl_car_body;

typedef struct crl_car_state {   /* one car: Car (car_dynamics.py:55-129) + its env bookkeeping */
    crl_car_body hull, wheel[4];
    float imp[4][3], motor_imp[4], motor_speed[4]; /* revolute joints hull<->wheel          */
    int32_t limit_state[4];
    double gas[4], omega[4], phase[4];             /* wheel attributes                       */
    double reward, prev_reward;                    /* CarRacing.rewards / prev_rewards       */
    int32_t tile_visited_count, last_block, done, step_count, first_step, pad;
    uint32_t wheel_tiles[4][CRL_CAR_MAX_TILES / 32]; /* w.tiles as bit sets                  */
    uint32_t visited[CRL_CAR_MAX_TILES / 32];        /* tile.road_visited[car]               */
    float sleep_time[5], pad2;                       /* b2Body::m_sleepTime: hull, wheels 0-3 */
} crl_car_state;

#define CRL_CAR_MAX_CONTACTS 8
typedef struct crl_car_contact { /* a touching contact between a fixture of car 0 and one of car 1 */
    int32_t pair;           /* fa * 8 + fb; fixtures 0-3 hull polygons, 4-7 wheels               */
    int32_t count, type;    /* manifold points (1-2); 0 = face of A is the reference, 1 = of B   */
    float ln[2], lp[2];     /* manifold local normal / point (reference body frame)              */
    float pt[2][2];         /* manifold points (incident body frame)                             */
    uint32_t id[2];         /* contact feature ids (warm-start key)                              */
    float nimp[2], timp[2]; /* accumulated normal / tangent impulses                             */
} crl_car_contact;

typedef struct crl_car_env_state {
    crl_car_state car[2];
    int32_t elapsed;  /* gym TimeLimit._elapsed_steps */
    uint32_t episode; /* resets so far                */
    int32_t n_contact;
    int32_t coupled; /* (get only) 1 = the step filed the env for the narrow phase: the cars' boxes met, or the cars touched in the step before */
    crl_car_contact contact[CRL_CAR_MAX_CONTACTS];
} crl_car_env_state;

int crl_car_get_state(crl_ctx *ctx, crl_car_env_state *state_host, int64_t first, int64_t count, void *stream);
int crl_car_set_state(crl_ctx *ctx, const crl_car_env_state *state_host, int64_t first, int64_t count, void *stream);
/* Track of env `env` as the physics keeps it: n tiles; tile_poly float32 [n][5][2] (counter-clockwise, as
 * b2PolygonShape stores them), border_poly float32 [n][4][2], border u8 [n] (0 none, 1 white, 2 red),
 * start_pose = track[0] (beta, x, y). */
int crl_car_get_track(crl_ctx *ctx, int64_t env, int32_t *n, float *tile_poly, float *border_poly, uint8_t *border,
                      float *start_pose, void *stream);
/* Replaces the track of env `env` (parity tests: a track built elsewhere).  The polygons are the reference's
 * road_poly entries in float64 and in its vertex order (_create_track, car_racing_multi_players.py:400-441):
 * tile i = (road1_l, road_m, road1_r, road2_r, road2_l), border quad (b1_l, b1_r, b2_r, b2_l) where border[i] != 0.
 * Derives the float32 sensors and re-rasters the env's observation map (render_road_for_observation_map, :732-755). */
int crl_car_set_track(crl_ctx *ctx, int64_t env, int32_t n, const double *tile_poly, const double *border_poly,
                      const uint8_t *border, const float *start_pose, void *stream);
/* The env's pre-rastered observation map, one palette index per pixel (0 grass, 1 lighter square, 2-4 road
 * 102 / 104 / 107, 5 white, 6 red): palette_host u8 [CRL_CAR_MAP_W][CRL_CAR_MAP_W]; *overflow (optional) = polygon
 * vertices that fell outside the window at the env's last reset (0 for every track). */
int crl_car_get_map(crl_ctx *ctx, int64_t env, uint8_t *palette_host, int32_t *overflow, void *stream);
/* How often a fixed capacity of the device state was hit since crl_create (synchronises `stream`): out4_host[0] = a wheel
 * touching more than 6 track tiles at once (w.tiles is a set in the reference, car_racing_multi_players.py:111-153; the extra
 * tile is not recorded), [1] = more than CRL_CAR_MAX_CONTACTS manifolds between the two cars of an env (the rest are dropped),
 * [2], [3] reserved.  Zero on every track of the tests and of a 10^7 env-step soak: a non-zero count means a deviation. */
int crl_car_cap_hits(crl_ctx *ctx, int32_t *out4_host, void *stream);
/* Replay mode for CarRacing.reset's randomness: per env `attempts` rows of 24 uniforms (the
 * np_random.uniform draws of one _create_track attempt) and one birth-place swap bit each. */
int crl_car_set_replay(crl_ctx *ctx, const double *u_host, const uint8_t *swap_host, int64_t attempts);

/* CarRacing.step's per-car returns that the VecEnv folds away, as device arrays valid until the
 * next step: done_car u8 (N, players) = the `done` dict (car_racing_multi_players.py:618),
 * num_steps i32 (N) = info[k]["num_steps"] = CarRacing.step_count after the step (:616-620),
 * captured before the auto-reset. */
int crl_car_info(crl_ctx *ctx, const uint8_t **done_car_dev, const int32_t **num_steps_dev);
/* Same data copied (device to device, on `stream`) into caller-owned arrays, plus elapsed i32 (N) = gym TimeLimit's
 * _elapsed_steps after the step, before the auto-reset (info["TimeLimit.truncated"] is set on the step where it reaches
 * max_episode_steps = 1000, car_racing/register.py:15-26); any destination may be NULL. */
int crl_car_copy_info(crl_ctx *ctx, uint8_t *done_car_out_dev, int32_t *num_steps_out_dev, int32_t *elapsed_out_dev, void *stream);

/* ---- FrameStackTensor.update on device (utils/utils.py:158-170) -----------------
 * stack f32 (N, C*k, H, W), in place:  stack *= mask[n];  planes shift down by C (roll -C on dim 1);
 * the last C planes = obs.  One pass over the tensor (torch's mul + roll + slice-assign are three).
 * obs: (N, C, H, W) with `obs_env_stride` ELEMENTS between envs (a view of a wider tensor is fine),
 * u8 or f32 per `obs_dtype` (crl_obs_dtype); mask: f32 (N) or NULL (= all ones).  hw = H*W. */
int crl_frame_stack_update(float *stack_dev, const void *obs_dev, int32_t obs_dtype, int64_t obs_env_stride,
                           const float *mask_dev, int64_t n, int32_t c, int32_t k, int64_t hw, void *stream);
/* The same update OUT of place: dst = (src * mask) shifted, with obs as the newest planes; src is not written.  This is the
 * reference's own data flow -- `self.current_obs = self.current_obs.roll(...)` (utils/utils.py:166-167) binds a NEW tensor on every
 * update -- and a plain streaming copy for the memory system (no load that must stay ahead of a store to the same rows).  dst and
 * src must not overlap (CRL_EINVAL). */
int crl_frame_stack_update_to(float *dst_stack_dev, const float *src_stack_dev, const void *obs_dev, int32_t obs_dtype, int64_t obs_env_stride,
                              const float *mask_dev, int64_t n, int32_t c, int32_t k, int64_t hw, void *stream);
/* The same update of a UINT8 stack (N, C*k, H, W) -- `FrameStackTensor(..., dtype=torch.uint8)`, an opt-in beside the reference's float32
 * contract (utils/utils.py:145-157 allocates float32) --, in place (dst == src) or into another tensor (no overlap: CRL_EINVAL).  Bytes
 * cannot be scaled: a mask entry of 0 erases the env's history, any other value keeps it (the masks of step_envs are 1 - done,
 * utils/utils.py:55-57); a float32 observation holds 0..255 integers and is truncated as a tensor cast would. */
int crl_frame_stack_update_u8(uint8_t *dst_stack_dev, const uint8_t *src_stack_dev, const void *obs_dev, int32_t obs_dtype, int64_t obs_env_stride,
                              const float *mask_dev, int64_t n, int32_t c, int32_t k, int64_t hw, void *stream);

/* ---- built-in CNN opponents of cPongTournament-v0 ----------------------------
 * Stands in for utils/policy_serving.py:10-66 `Policy(..., use_light_model=True)` as built by
 * pong/builtin_policies.py:61-91 for WEAK / MEDIUM: LightActorCritic (utils/network.py:73-93:
 * x/255 -> conv 4->16 k4 s2 -> ReLU -> conv 16->16 k2 s2 -> ReLU -> 1600 -> 3 logits) on the
 * policy's OWN stack of the last four 42x42 frames it was shown (FrameStackTensor.update without
 * a mask, utils/utils.py:159-170: never cleared at episode ends), action = argmax of the logits.
 * Weights are the model tensors of the checkpoint, float32, in torch layout. */
#define CRL_POLICY_DIM 42
#define CRL_POLICY_STACK 4
typedef struct crl_policy crl_policy;
int crl_policy_create(int32_t device, int64_t num_envs, const float *conv1_w_host /*[16,4,4,4]*/,
                      const float *conv1_b_host /*[16]*/, const float *conv2_w_host /*[16,16,2,2]*/,
                      const float *conv2_b_host /*[16]*/, const float *actor_w_host /*[3,1600]*/,
                      const float *actor_b_host /*[3]*/, crl_policy **out);
/* Policy(..., use_light_model=False) (policy_serving.py:21-25): ActorCritic (utils/network.py:14-50: conv 4->16 k4 s2, conv 16->32 k4
 * s2 pad 2, conv 32->256 k11, actor 256->3) -- the model of the reference's STRONG / ALPHA_PONG opponents -- behind the same
 * handle: crl_policy_reset / _act / _get_stack / _set_stack / _destroy work on it unchanged. */
int crl_policy_create_full(int32_t device, int64_t num_envs, const float *conv1_w_host /*[16,4,4,4]*/,
                           const float *conv1_b_host /*[16]*/, const float *conv2_w_host /*[32,16,4,4]*/,
                           const float *conv2_b_host /*[32]*/, const float *conv3_w_host /*[256,32,11,11]*/,
                           const float *conv3_b_host /*[256]*/, const float *actor_w_host /*[3,256]*/,
                           const float *actor_b_host /*[3]*/, crl_policy **out);
void crl_policy_destroy(crl_policy *p);
/* Policy.reset (policy_serving.py:46-47): zero the frame stack. */
int crl_policy_reset(crl_policy *p, void *stream);
/* Policy.__call__ (policy_serving.py:58-66): push one 42x42 u8 frame per env (env i at
 * frame_dev + i * frame_stride bytes; stride a multiple of 4) onto the stack and write the
 * greedy action of env i to actions_dev[i * action_stride] (int32 elements; pass 2 to fill the
 * right-hand column of an (N, 2) cPongDouble action array in place).  logits_dev: optional
 * float32 [N, 3]. */
int crl_policy_act(crl_policy *p, const uint8_t *frame_dev, int64_t frame_stride, int32_t *actions_dev,
                   int64_t action_stride, float *logits_dev, void *stream);
/* The stack as the model sees it: u8 [N, 4, 42, 42], oldest plane first (tests, checkpoints). */
int crl_policy_get_stack(crl_policy *p, uint8_t *stack_out_dev, void *stream);
int crl_policy_set_stack(crl_policy *p, const uint8_t *stack_in_dev, void *stream);

/* Text of the most recent failing call: of the calling thread (any call, crl_create included), or of one context. */
const char *crl_last_error(void);
const char *crl_ctx_last_error(const crl_ctx *ctx);
const char *crl_version(void);

/* Device self-test of include/crl_rot.h (no context needed): evaluates crl_sincosf on the float32 bit patterns
 * [first_bits, first_bits + count) that lie in its domain and compares each result with the double-double evaluation of
 * include/crl_f64.h rounded once to float32.  out5_host: [0] arguments tested, [1] sine results that are not the correctly
 * rounded float32, [2] cosine likewise, [3] results too close to a rounding boundary for the double-double bound to decide,
 * [4] first mismatching bit pattern + 1 (0 = none).  The whole space (first_bits 0, count 2^32) takes a few seconds.
 * Replaces nothing in the reference: Box2D calls the host libm's sinf / cosf (b2Math.h b2Rot::Set, via box2d-py,
 * car_racing/car_racing_multi_players.py:600); this pins the one evaluation both sides of the parity tests share. */
int crl_selftest_sincosf(int32_t device, uint64_t first_bits, uint64_t count, uint64_t *out5_host);

#ifdef __cplusplus
}
#endif
#endif /* CRL_H_ */
