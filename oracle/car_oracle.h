/* car_oracle.h -- types of the CarRacing CPU oracle (TEST INFRASTRUCTURE ONLY). */
#ifndef CAR_ORACLE_H_
#define CAR_ORACLE_H_
#include <stdint.h>

#define CAR_MAX_TILES 512

typedef struct car_track {
    int32_t n, pad;
    double track[CAR_MAX_TILES][4];          /* alpha, beta, x, y            */
    double tile[CAR_MAX_TILES][5][2];        /* road1_l, road_m, road1_r, road2_r, road2_l */
    double border_poly[CAR_MAX_TILES][4][2]; /* b1_l, b1_r, b2_r, b2_l (valid where border) */
    uint8_t border[CAR_MAX_TILES];
} car_track;

typedef struct car_body { /* b2Body: sweep.c, sweep.a, linear/angular velocity, force */
    float cx, cy, a, vx, vy, w, fx, fy;
} car_body;

typedef struct car_state {
    car_body hull, wheel[4];
    float imp[4][3], motor_imp[4], motor_speed[4]; /* b2RevoluteJoint m_impulse, m_motorImpulse, m_motorSpeed */
    int32_t limit_state[4];
    double gas[4], brake[4], steer[4], phase[4], omega[4]; /* Car wheel attributes (python floats) */
    float sleep_time[5], pad_s;                            /* b2Body::m_sleepTime of hull, wheels 0-3 */
} car_state;

typedef struct car_consts {
    float hull_poly[4][8][2];
    int32_t hull_n[4];
    float wheel_poly[4][2];
    float hull_mass, hull_inv_mass, hull_I, hull_inv_I, hull_lc[2];
    float wheel_mass, wheel_inv_mass, wheel_I, wheel_inv_I;
    float anchor[4][2];
} car_consts;

#define CAR_MAX_CONTACTS 8
typedef struct car_contact { /* one touching b2Contact between a fixture of car 0 and one of car 1 */
    int32_t pair;        /* fa * 8 + fb; fixtures 0-3 = hull polygons, 4-7 = wheels        */
    int32_t count, type; /* manifold points (1-2); 0 = e_faceA, 1 = e_faceB                 */
    float ln[2], lp[2];  /* manifold.localNormal / localPoint (reference body frame)       */
    float pt[2][2];      /* manifold.points[i].localPoint (incident body frame)            */
    uint32_t id[2];      /* b2ContactID key                                                */
    float nimp[2], timp[2]; /* accumulated normal / tangent impulses (warm start)          */
} car_contact;

typedef struct car_env {
    car_track trk;
    float tile32[CAR_MAX_TILES][5][2]; /* CCW, float32 as b2PolygonShape stores them */
    float tile_aabb[CAR_MAX_TILES][4];
    car_state car[2];
    uint32_t wheel_tiles[2][4][CAR_MAX_TILES / 32]; /* w.tiles            */
    uint32_t visited[2][CAR_MAX_TILES / 32];        /* tile.road_visited   */
    int32_t tile_visited_count[2], last_block[2], done[2];
    double reward[2], prev_reward[2], t;
    int32_t step_count;
    float inv_dt0;
    int32_t n_contact, contacts_enabled;
    car_contact contact[CAR_MAX_CONTACTS];
} car_env;

int car_oracle_create_track(const double *u, car_track *out);
const car_consts *car_oracle_consts(void);
void car_oracle_place(car_state *car, double init_angle, double init_x, double init_y, int birth_place_index);
void car_oracle_process_action(const double a[2], double out[3]);
void car_oracle_controls(car_state *car, double steer, double gas, double brake);
void car_oracle_wheel(double dt, double steer, double gas, double brake, double joint_angle, double q_sin, double q_cos,
                      double vx, double vy, int on_road, double *omega, double *phase, double *motor_speed, double force[2]);
int car_oracle_reset(car_env *e, const double *u, int max_attempts, int shuffle_swap);
void car_oracle_step(car_env *e, const double (*actions)[2], double step_reward[2], int done[2]);
void car_oracle_step_repeat(car_env *e, const double (*actions)[2], int repeat, double step_reward[2], int done[2]);
void car_oracle_contact_event(car_env *e, int c, int w, int t, int begin);
int car_oracle_wheel_on_road(const car_env *e, int c, int w);
void car_oracle_hull_position(const car_env *e, int c, float out[3]);
int car_oracle_env_size(void);
/* observation: palette map of the window [org, org + w)^2 of the 10000^2 surface, then the view */
#define CAR_MAP_ORG 4392
#define CAR_MAP_W 1216
long car_oracle_build_map(const car_env *e, uint8_t *map, int org, int w);
void car_oracle_map_vertices(const car_env *e, int32_t *out);
void car_oracle_view_sources(const car_env *e, int viewer, int32_t *src_xy, int32_t *rect_xy);
void car_oracle_render(const car_env *e, const uint8_t *map, int org, int w, int viewer, uint8_t *out);
void car_oracle_render_analytic(const car_env *e, int viewer, uint8_t *out);
void car_oracle_set_text(const uint32_t *bits);
void car_oracle_f64(int fn, const double *a, const double *b, double *out, long n);
#endif
