// Swarm-v0 on gfx950: SwarmEnv._step / _reset (reference fed_gym/envs/multiagent.py:30-115),
// gym TimeLimit, and SwarmStateProcessor.process_state (fed_gym/agents/state_processors.py:17-42)
// as hand-written HIP.
//
// Mapping: one lane per locust, 4 envs per 320-lane workgroup (5 full waves, no idle lanes);
// the 90 source points of each env (80 locusts + 10 agents, float64 x/y interleaved) live in LDS
// and are read as wave-broadcast ds_read_b128; each lane runs the 90-term force sum for its own
// locust in the reference's (numpy pairwise) association order, so no cross-lane reduction is
// needed for v.  Agents are advanced by the first 40 lanes.  The observation (bin indices) is
// produced by the same launch from the positions already in LDS/registers; the dense 84x84 image
// is never written.  HBM traffic per env-step: x 1280 + xa 160 + noise rows 1440 +
// actions 80 read; x 1280 + xa 160 + bins 200 + reward/done/elapsed ~17 written.
#include "common.h"
#include "rng.h"

namespace grl {

struct SwarmParams {
    double *x, *xa, *pnoise, *anoise;
    const double *rx, *rxa, *rpn, *ran;               // reset snapshot
    const double *inj_x0, *inj_xa0, *inj_ra, *inj_an, *inj_pn;  // injected reset draws
    const float *actions;
    const double *actions64;     // if set: float64 actions (SwarmEnv.step called directly, e.g. by the eval monitor) instead of `actions`
    int32_t *elapsed, *episode;
    float *reward;
    double *reward64;
    uint8_t *done, *lbins, *abins, *pos;
    int32_t *done_list, *done_count;
    const int32_t *reset_list, *reset_count;
    int E, grid, max_steps;
    int env_base;                // MODE_STEP / MODE_OBSERVE: the launch covers envs [env_base, E) (E = end of the range)
    uint32_t flags, env_off;
    uint64_t seed;
    int no_wind;                 // MODE_STEP: SwarmEnv._step(v_action, add_wind=False) -- the agents' action row is used as it is (multiagent.py:33-36)
};

constexpr int LDS_STRIDE = 97;   // 90 points padded so two envs' rows start on different banks

struct SwarmLds {
    double2 p[SWARM_EPB][LDS_STRIDE];
    double en[SWARM_EPB][N_LOCUSTS];
    double meanx[SWARM_EPB];
    double rew[SWARM_EPB];
    int dflag[SWARM_EPB];      // env id of a finished episode (or -1): folded into ONE atomicAdd per workgroup
};

constexpr double DT = 0.05, NOISE = 0.0001, WIND = 1.0, GRAV = -1.0, FATT = 0.5, LATT = 10.0;

// xv_cutoff (multiagent.py:77-86)
__device__ __forceinline__ void cutoff(double &y, double &vx, double &vy) {
    if (y <= 0) {
        y = 0;
        vx = 0;
        if (vy <= 0) vy = 0;
    }
}

// x_update (multiagent.py:70-75); n* are raw N(0,1) draws
__device__ __forceinline__ void x_update(double &x, double &y, double vx, double vy, double nx, double ny) {
    cutoff(y, vx, vy);
    x = x + (DT * vx + NOISE * nx);
    y = y + (DT * vy + NOISE * ny);
    if (y <= 0) y = 0;
}

// x_update for the agents when the action row is FLOAT32 -- what the worker reads from the learner's shared c_float array
// (quirk Q7; emulator_runner.py:126, paac.py:269).  numpy keeps the row's dtype through `v_action[:, 0] += WIND_SPEED`
// (multiagent.py:33-36) and through `dt * v` (a Python float times a float32 array is a float32 product with dt rounded to
// float32, multiagent.py:72); only the sum with the float64 noise is float64.  Pinned by tests/golden/swarm_runner.npz
// (SwarmRunner._run itself): evaluating the action term in float64 differs by ~6e-9 per step, which the chaotic dynamics amplify.
__device__ __forceinline__ void x_update_f32v(double &x, double &y, float vx, float vy, double nx, double ny) {
    if (y <= 0) {
        y = 0;
        vx = 0;
        if (vy <= 0) vy = 0;
    }
    x = x + ((double)(0.05f * vx) + NOISE * nx);
    y = y + ((double)(0.05f * vy) + NOISE * ny);
    if (y <= 0) y = 0;
}

template <bool FAST>
__device__ __forceinline__ void pair_term(double sx, double sy, double xj, double yj, double &t0, double &t1) {
    double dx = sx - xj, dy = sy - yj;
    double d = sqrt(dx * dx + dy * dy);
    if (FAST) {
        double t = exp(d * -0.1);
        double t2 = t * t, t4 = t2 * t2, t5 = t4 * t, t10 = t5 * t5;
        double w = (FATT * t - t10) / (d + 0.000001);
        t0 = w * dx;
        t1 = w * dy;
    } else {
        // s(r) = F*exp(-r/L) - exp(-r)   (multiagent.py:65-68); term = s*dx/(r+1e-6) (:103-104)
        double s = FATT * exp(-d / LATT) - exp(-d);
        double den = d + 0.000001;
        t0 = s * dx / den;
        t1 = s * dy / den;
    }
}

// v_calculate for one target locust (multiagent.py:100-113).  Sums follow numpy's pairwise
// order for n=80 and n=10: 8 strided accumulators, a fixed tree, then the tail.
template <bool FAST>
__device__ __forceinline__ void locust_velocity(const double2 *src, double xj, double yj, double &vx, double &vy) {
    double a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pair_term<FAST>(src[k].x, src[k].y, xj, yj, a0[k], a1[k]);
    for (int i = 8; i < N_LOCUSTS; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double t0, t1;
            pair_term<FAST>(src[i + k].x, src[i + k].y, xj, yj, t0, t1);
            a0[k] += t0;
            a1[k] += t1;
        }
    }
    double ll0 = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7]));
    double ll1 = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
#pragma unroll
    for (int k = 0; k < 8; ++k) pair_term<FAST>(src[N_LOCUSTS + k].x, src[N_LOCUSTS + k].y, xj, yj, a0[k], a1[k]);
    double al0 = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7]));
    double al1 = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
#pragma unroll
    for (int k = 8; k < N_AGENTS; ++k) {
        double t0, t1;
        pair_term<FAST>(src[N_LOCUSTS + k].x, src[N_LOCUSTS + k].y, xj, yj, t0, t1);
        al0 += t0;
        al1 += t1;
    }
    vx = (WIND + ll0) + al0;
    vy = (GRAV + ll1) + al1;
}

// One SwarmEnv._step for the 4 envs of the block, state in LDS (agents) / registers (own locust).
// Agent lanes (tid < 40) carry (actx, acty) = action BEFORE wind and (anx, any) raw noise; act32: the action came from a
// float32 row and its wind / dt arithmetic is float32 (x_update_f32v).
// On return L.p holds the new positions of all 90 points, L.rew[el] the reward; ends on a barrier.
template <bool FAST>
__device__ __forceinline__ void block_step(SwarmLds &L, int tid, int el, int j, double &xj, double &yj, double actx,
                                           double acty, double anx, double any, double pnx, double pny, bool act32 = false,
                                           bool wind = true) {
    if (tid < SWARM_EPB * N_AGENTS) {
        int ea = tid / N_AGENTS, a = tid - ea * N_AGENTS;
        double2 q = L.p[ea][N_LOCUSTS + a];
        // wind: `if add_wind: v_action[:, 0] += WIND_SPEED` (multiagent.py:35-36); the locusts' U in v_calculate does not depend on it
        if (act32) x_update_f32v(q.x, q.y, wind ? (float)actx + 1.0f : (float)actx, (float)acty, anx, any);
        else x_update(q.x, q.y, wind ? actx + WIND : actx, acty, anx, any);
        L.p[ea][N_LOCUSTS + a] = q;
    }
    L.p[el][j] = make_double2(xj, yj);
    __syncthreads();
    double vx, vy;
    locust_velocity<FAST>(L.p[el], xj, yj, vx, vy);
    L.en[el][j] = vx * vx + vy * vy;
    x_update(xj, yj, vx, vy, pnx, pny);
    __syncthreads();
    L.p[el][j] = make_double2(xj, yj);
    if (j == 0) {   // energy = mean_j |v_j|^2, numpy pairwise order for n=80 (multiagent.py:114)
        const double *e = L.en[el];
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = e[k];
        for (int i = 8; i < N_LOCUSTS; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += e[i + k];
        }
        double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        L.rew[el] = -(s / (double)N_LOCUSTS);
    }
    __syncthreads();
}

// searchsorted(edges, v, 'right') for edges = linspace(lo, hi, G+1) as numpy builds them:
// edge(k) = k*step + lo (two roundings), edge(G) = hi exactly (state_processors.py:31 via histogramdd).
__device__ __forceinline__ double edge_at(int k, double lo, double hi, double step, int G) {
    return (k >= G) ? hi : ((double)k * step + lo);
}
__device__ __forceinline__ int count_edges_le(double v, double lo, double hi, double step, int G) {
    if (!(v >= lo)) return 0;        // below the first edge (or NaN)
    if (v >= hi) return G + 1;
    double gf = floor((v - lo) / step);
    int g = (gf > (double)G) ? G : (int)gf;   // guess: index of the last edge <= v
    if (g < 0) g = 0;
    for (int it = 0; it < 4 && g < G && edge_at(g + 1, lo, hi, step, G) <= v; ++it) ++g;
    for (int it = 0; it < 4 && g > 0 && edge_at(g, lo, hi, step, G) > v; ++it) --g;
    return g + 1;
}

enum { MODE_STEP = 0, MODE_OBSERVE = 1, MODE_RESET = 2 };

template <int MODE, bool FAST>
// The step's pair loop is a long chain of dependent fp64 operations (two exp polynomials, three divisions per pair): five waves per SIMD
// instead of the four its 124 registers allowed hide more of it -- 0.976 -> 0.875 ms per 32 768-env step with 96 registers and 24 bytes of
// scratch per lane (round 4).  Six waves (80 registers, 104 bytes of scratch) measured 0.859 ms but tripled the kernel's HBM-side bytes
// (each lane writes and reads its spill slots once: 488 MB per launch against 151 MB algorithmic); 7 / 8 waves: 0.875 / 0.863 ms.
// The second launch bound is waves per SIMD.
__global__ __launch_bounds__(SWARM_TPB, MODE == 0 ? 5 : 1) void swarm_kernel(SwarmParams P) {
    __shared__ SwarmLds L;
    __shared__ int env_of[SWARM_EPB];
    const int tid = threadIdx.x;
    const int el = tid / N_LOCUSTS, j = tid - el * N_LOCUSTS;

    if (MODE == MODE_RESET) {
        int cnt = *P.reset_count;
        if ((int)blockIdx.x * SWARM_EPB >= cnt) return;   // block-uniform
        if (tid < SWARM_EPB) {
            int li = blockIdx.x * SWARM_EPB + tid;
            env_of[tid] = li < cnt ? P.reset_list[li] : -1;
        }
    } else {
        if (tid < SWARM_EPB) {
            int e = P.env_base + blockIdx.x * SWARM_EPB + tid;
            env_of[tid] = e < P.E ? e : -1;
        }
    }
    __syncthreads();
    const int env = env_of[el];
    const bool active = env >= 0;
    const int ea = tid / N_AGENTS, a = tid - ea * N_AGENTS;   // agent-lane view (valid when tid < 40)
    const bool agent_lane = tid < SWARM_EPB * N_AGENTS;
    const int aenv = agent_lane ? env_of[ea] : -1;
    const bool aactive = aenv >= 0;

    double xj = 0.5, yj = 0.5;
    if (MODE == MODE_STEP || MODE == MODE_OBSERVE) {
        if (active) {
            double2 q = reinterpret_cast<const double2 *>(P.x)[(size_t)env * N_LOCUSTS + j];
            xj = q.x; yj = q.y;
        }
        if (agent_lane) {
            double2 q = make_double2(0.5, 0.5);
            if (aactive) q = reinterpret_cast<const double2 *>(P.xa)[(size_t)aenv * N_AGENTS + a];
            L.p[ea][N_LOCUSTS + a] = q;
        }
    }

    if (MODE == MODE_STEP) {
        double actx = 0, acty = 0, anx = 0, any = 0, pnx = 0, pny = 0;
        if (active) {
            double2 n = reinterpret_cast<const double2 *>(P.pnoise)[(size_t)env * N_LOCUSTS + j];
            pnx = n.x; pny = n.y;
        }
        if (aactive) {
            if (P.actions64) {
                double2 act = reinterpret_cast<const double2 *>(P.actions64)[(size_t)aenv * N_AGENTS + a];
                actx = act.x; acty = act.y;
            } else {
                float2 act = reinterpret_cast<const float2 *>(P.actions)[(size_t)aenv * N_AGENTS + a];
                actx = (double)act.x; acty = (double)act.y;
            }
            double2 n = reinterpret_cast<const double2 *>(P.anoise)[(size_t)aenv * N_AGENTS + a];
            anx = n.x; any = n.y;
        }
        block_step<FAST>(L, tid, el, j, xj, yj, actx, acty, anx, any, pnx, pny, P.actions64 == nullptr, P.no_wind == 0);
        if (active) reinterpret_cast<double2 *>(P.x)[(size_t)env * N_LOCUSTS + j] = make_double2(xj, yj);
        if (aactive) reinterpret_cast<double2 *>(P.xa)[(size_t)aenv * N_AGENTS + a] = L.p[ea][N_LOCUSTS + a];
        if (active && j == 0) {
            double r = L.rew[el];
            int el_new = P.elapsed[env] + 1;            // gym TimeLimit: counts wrapped steps
            bool d = (r >= 0) || (P.max_steps > 0 && el_new >= P.max_steps);   // multiagent.py:44 + TimeLimit
            P.elapsed[env] = el_new;
            P.reward64[env] = r;
            P.reward[env] = (float)r;
            P.done[env] = d ? 1 : 0;
            L.dflag[el] = d ? env : -1;
        } else if (j == 0) {
            L.dflag[el] = -1;
        }
        // episode-done compaction (emulator_runner.py:128-132 resets exactly these): the workgroup's finished envs take ONE slot
        // range with one atomicAdd (at a TimeLimit boundary every env of the batch finishes in the same launch); the order of
        // the list across workgroups is arrival order, grl_read_output("done_list") sorts it
        __syncthreads();
        if (tid == 0) {
            int n = 0;
#pragma unroll
            for (int e = 0; e < SWARM_EPB; ++e) n += L.dflag[e] >= 0 ? 1 : 0;
            if (n) {
                int slot = atomicAdd(P.done_count, n);
#pragma unroll
                for (int e = 0; e < SWARM_EPB; ++e)
                    if (L.dflag[e] >= 0) P.done_list[slot++] = L.dflag[e];
            }
        }
    }

    if (MODE == MODE_RESET) {
        const uint32_t genv = active ? (uint32_t)env + P.env_off : 0u;
        const uint32_t gaenv = aactive ? (uint32_t)aenv + P.env_off : 0u;
        const uint32_t ep = (active && !(P.flags & GRL_F_RESEED_EACH_RESET)) ? (uint32_t)P.episode[env] : 0u;
        const uint32_t aep = (aactive && !(P.flags & GRL_F_RESEED_EACH_RESET)) ? (uint32_t)P.episode[aenv] : 0u;
        double pn10x = 0, pn10y = 0, an10x = 0, an10y = 0;
        if (P.flags & GRL_F_RESET_FROM_SNAPSHOT) {
            if (active) {
                double2 q = reinterpret_cast<const double2 *>(P.rx)[(size_t)env * N_LOCUSTS + j];
                xj = q.x; yj = q.y;
                double2 n = reinterpret_cast<const double2 *>(P.rpn)[(size_t)env * N_LOCUSTS + j];
                pn10x = n.x; pn10y = n.y;
            }
            if (agent_lane) {
                double2 q = make_double2(0.5, 0.5);
                if (aactive) {
                    q = reinterpret_cast<const double2 *>(P.rxa)[(size_t)aenv * N_AGENTS + a];
                    double2 n = reinterpret_cast<const double2 *>(P.ran)[(size_t)aenv * N_AGENTS + a];
                    an10x = n.x; an10y = n.y;
                }
                L.p[ea][N_LOCUSTS + a] = q;
            }
            L.p[el][j] = make_double2(xj, yj);
            __syncthreads();
        } else {
            const bool inj = P.inj_x0 != nullptr;
            // x ~ U[0,1)^(80,2), xa ~ U[0,1)^(10,2)  (multiagent.py:51-52)
            if (inj) {
                if (active) {
                    double2 q = reinterpret_cast<const double2 *>(P.inj_x0)[(size_t)env * N_LOCUSTS + j];
                    xj = q.x; yj = q.y;
                }
            } else {
                u01_pair(rng_block(P.seed, genv, ep, RS_SWARM_X0, j), xj, yj);
            }
            if (agent_lane) {
                double2 q = make_double2(0.5, 0.5);
                if (inj) {
                    if (aactive) q = reinterpret_cast<const double2 *>(P.inj_xa0)[(size_t)aenv * N_AGENTS + a];
                } else {
                    u01_pair(rng_block(P.seed, gaenv, aep, RS_SWARM_XA0, a), q.x, q.y);
                }
                L.p[ea][N_LOCUSTS + a] = q;
            }
            // N_BURN_IN steps with noise rows 0..9 (multiagent.py:58-61), then row 10 is kept (quirk Q1)
            for (int t = 0; t <= N_BURN_IN; ++t) {
                double actx = 0, acty = 0, anx = 0, any = 0, pnx = 0, pny = 0;
                if (inj) {
                    if (active) {
                        double2 n = reinterpret_cast<const double2 *>(P.inj_pn)[((size_t)env * (N_BURN_IN + 1) + t) * N_LOCUSTS + j];
                        pnx = n.x; pny = n.y;
                    }
                    if (aactive) {
                        double2 n = reinterpret_cast<const double2 *>(P.inj_an)[((size_t)aenv * (N_BURN_IN + 1) + t) * N_AGENTS + a];
                        anx = n.x; any = n.y;
                        if (t < N_BURN_IN) {
                            double2 r = reinterpret_cast<const double2 *>(P.inj_ra)[((size_t)aenv * N_BURN_IN + t) * N_AGENTS + a];
                            actx = r.x; acty = r.y;
                        }
                    }
                } else {
                    normal_pair(rng_block(P.seed, genv, ep, RS_SWARM_PNOISE, t * N_LOCUSTS + j), pnx, pny);
                    if (agent_lane) {
                        normal_pair(rng_block(P.seed, gaenv, aep, RS_SWARM_ANOISE, t * N_AGENTS + a), anx, any);
                        if (t < N_BURN_IN) normal_pair(rng_block(P.seed, gaenv, aep, RS_SWARM_RANDACT, t * N_AGENTS + a), actx, acty);
                    }
                }
                if (t == N_BURN_IN) {
                    pn10x = pnx; pn10y = pny; an10x = anx; an10y = any;
                    break;
                }
                block_step<FAST>(L, tid, el, j, xj, yj, actx, acty, anx, any, pnx, pny);
            }
        }
        if (active) {
            reinterpret_cast<double2 *>(P.x)[(size_t)env * N_LOCUSTS + j] = make_double2(xj, yj);
            reinterpret_cast<double2 *>(P.pnoise)[(size_t)env * N_LOCUSTS + j] = make_double2(pn10x, pn10y);
            if (j == 0) {
                P.elapsed[env] = 0;
                P.episode[env] = P.episode[env] + 1;
            }
        }
        if (aactive) {
            reinterpret_cast<double2 *>(P.xa)[(size_t)aenv * N_AGENTS + a] = L.p[ea][N_LOCUSTS + a];
            reinterpret_cast<double2 *>(P.anoise)[(size_t)aenv * N_AGENTS + a] = make_double2(an10x, an10y);
        }
    }

    if (MODE == MODE_OBSERVE) {
        L.p[el][j] = make_double2(xj, yj);
        __syncthreads();
    }

    // ---- observation: process_state (state_processors.py:29-42) -------------------------------
    if (P.flags & GRL_F_SWARM_NO_OBSERVE) return;
    if (j == 0) {   // np.mean(vstack([x, xa]), axis=0)[0]: sequential row-by-row sum, then /90
        double s = L.p[el][0].x;
        for (int i = 1; i < N_POINTS; ++i) s += L.p[el][i].x;
        L.meanx[el] = s / (double)N_POINTS;
    }
    __syncthreads();
    const int G = P.grid;
    const double ylo = 0.0, yhi = 6.0;                 // [0, 2*HEIGHT]  (state_processors.py:27)
    const double ystep = (yhi - ylo) / (double)G;
    {
        double m = L.meanx[el];
        double lo = m - 1.5, hi = m + 1.5;             // WIDTH/2 (state_processors.py:27)
        double step = (hi - lo) / (double)G;
        int cx = count_edges_le(xj, lo, hi, step, G);
        int cy = count_edges_le(yj, ylo, yhi, ystep, G);
        int bx = (xj == hi) ? G - 1 : cx - 1;          // histogramdd: a value on the last edge joins the last bin
        int by = (yj == yhi) ? G - 1 : cy - 1;
        bool in = bx >= 0 && bx < G && by >= 0 && by < G;
        if (active) reinterpret_cast<uint16_t *>(P.lbins)[(size_t)env * N_LOCUSTS + j] = in ? (uint16_t)(bx | (by << 8)) : (uint16_t)0xFFFF;
    }
    if (aactive) {
        double2 q = L.p[ea][N_LOCUSTS + a];
        double m = L.meanx[ea];
        double lo = m - 1.5, hi = m + 1.5;
        double step = (hi - lo) / (double)G;
        int cx = count_edges_le(q.x, lo, hi, step, G);
        int cy = count_edges_le(q.y, ylo, yhi, ystep, G);
        int bx = (q.x == hi) ? G - 1 : cx - 1;
        int by = (q.y == yhi) ? G - 1 : cy - 1;
        bool in = bx >= 0 && bx < G && by >= 0 && by < G;
        reinterpret_cast<uint16_t *>(P.abins)[(size_t)aenv * N_AGENTS + a] = in ? (uint16_t)(bx | (by << 8)) : (uint16_t)0xFFFF;
        // np.digitize == count of edges <= v, clamped to G-1 (state_processors.py:35-40, quirk Q2)
        int px = cx >= G ? G - 1 : cx;
        int py = cy >= G ? G - 1 : cy;
        reinterpret_cast<uint16_t *>(P.pos)[(size_t)aenv * N_AGENTS + a] = (uint16_t)(px | (py << 8));
    }
}

// get_local_states (paac/emulator_runner.py:98-111) as a dense f32 tensor; compat/debug only.
__global__ void swarm_materialize_kernel(const uint8_t *lbins, const uint8_t *abins, const uint8_t *pos, int first, int G,
                                         float *out) {
    extern __shared__ unsigned int cnt[];   // [2][G*G]
    const int env = first + blockIdx.x;
    const int GG = G * G;
    for (int i = threadIdx.x; i < 2 * GG; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    if (threadIdx.x < N_LOCUSTS) {
        int bx = lbins[((size_t)env * N_LOCUSTS + threadIdx.x) * 2], by = lbins[((size_t)env * N_LOCUSTS + threadIdx.x) * 2 + 1];
        if (bx != 255) atomicAdd(&cnt[bx * G + by], 1u);
    } else if (threadIdx.x < N_POINTS) {
        int a = threadIdx.x - N_LOCUSTS;
        int bx = abins[((size_t)env * N_AGENTS + a) * 2], by = abins[((size_t)env * N_AGENTS + a) * 2 + 1];
        if (bx != 255) atomicAdd(&cnt[GG + bx * G + by], 1u);
    }
    __syncthreads();
    float *o = out + (size_t)blockIdx.x * N_AGENTS * GG * 3;
    for (int a = 0; a < N_AGENTS; ++a) {
        int px = pos[((size_t)env * N_AGENTS + a) * 2], py = pos[((size_t)env * N_AGENTS + a) * 2 + 1];
        int hot = px * G + py;
        for (int i = threadIdx.x; i < GG * 3; i += blockDim.x) {
            int pix = i / 3, c = i - pix * 3;
            float v;
            if (c == 0) v = (float)((double)cnt[pix] / 80.0);          // x_grid / len(state[0])
            else if (c == 1) v = (float)((double)cnt[GG + pix] / 10.0);  // xa_grid / len(state[1])
            else v = (pix == hot) ? 1.0f : 0.0f;
            o[(size_t)a * GG * 3 + i] = v;
        }
    }
}

static SwarmParams make_params(grl_handle *h) {
    SwarmParams P{};
    P.x = h->sw.x; P.xa = h->sw.xa; P.pnoise = h->sw.pnoise; P.anoise = h->sw.anoise;
    P.rx = h->sw.rx; P.rxa = h->sw.rxa; P.rpn = h->sw.rpnoise; P.ran = h->sw.ranoise;
    P.elapsed = h->elapsed; P.episode = h->episode; P.reward = h->reward; P.reward64 = h->sw.reward64;
    P.done = h->done; P.lbins = h->sw.lbins; P.abins = h->sw.abins; P.pos = h->sw.pos;
    P.done_list = h->done_list; P.done_count = h->done_count;
    P.E = h->E; P.grid = h->cfg.grid_size; P.max_steps = h->cfg.max_episode_steps;
    P.flags = h->cfg.flags; P.env_off = (uint32_t)h->cfg.env_id_offset; P.seed = h->cfg.seed;
    return P;
}

template <typename T>
static int dmalloc(grl_handle *h, T **p, size_t n) {
    GRL_HIP(h, hipMalloc((void **)p, n * sizeof(T)));
    h->allocs.push_back(*p);
    GRL_HIP(h, hipMemsetAsync(*p, 0, n * sizeof(T), h->stream));
    return GRL_OK;
}

int swarm_alloc(grl_handle *h) {
    size_t E = h->E;
    int rc;
    if ((rc = dmalloc(h, &h->sw.x, E * N_LOCUSTS * 2))) return rc;
    if ((rc = dmalloc(h, &h->sw.xa, E * N_AGENTS * 2))) return rc;
    if ((rc = dmalloc(h, &h->sw.pnoise, E * N_LOCUSTS * 2))) return rc;
    if ((rc = dmalloc(h, &h->sw.anoise, E * N_AGENTS * 2))) return rc;
    if ((rc = dmalloc(h, &h->sw.reward64, E))) return rc;
    if ((rc = dmalloc(h, &h->sw.lbins, E * N_LOCUSTS * 2))) return rc;
    if ((rc = dmalloc(h, &h->sw.abins, E * N_AGENTS * 2))) return rc;
    if ((rc = dmalloc(h, &h->sw.pos, E * N_AGENTS * 2))) return rc;
    if (h->cfg.flags & GRL_F_RESET_FROM_SNAPSHOT) {
        if ((rc = dmalloc(h, &h->sw.rx, E * N_LOCUSTS * 2))) return rc;
        if ((rc = dmalloc(h, &h->sw.rxa, E * N_AGENTS * 2))) return rc;
        if ((rc = dmalloc(h, &h->sw.rpnoise, E * N_LOCUSTS * 2))) return rc;
        if ((rc = dmalloc(h, &h->sw.ranoise, E * N_AGENTS * 2))) return rc;
    }
    return GRL_OK;
}

static inline int nblocks(int n) { return (n + SWARM_EPB - 1) / SWARM_EPB; }

int swarm_launch_step(grl_handle *h, const float *actions_dev, const double *actions64_dev, int no_wind) {
    SwarmParams P = make_params(h);
    P.actions = actions_dev;
    P.actions64 = actions64_dev;
    P.no_wind = no_wind;
    GRL_HIP(h, hipMemsetAsync(h->done_count, 0, sizeof(int32_t), h->stream));
    prof_begin(h);
    if (h->cfg.flags & GRL_F_SWARM_FAST_MATH)
        hipLaunchKernelGGL((swarm_kernel<MODE_STEP, true>), dim3(nblocks(h->E)), dim3(SWARM_TPB), 0, h->stream, P);
    else
        hipLaunchKernelGGL((swarm_kernel<MODE_STEP, false>), dim3(nblocks(h->E)), dim3(SWARM_TPB), 0, h->stream, P);
    prof_end(h);
    GRL_HIP(h, hipGetLastError());
    // auto-reset of the envs that just finished (paac/emulator_runner.py:128-132): the terminal
    // reward/done stay, the observation becomes the reset one (quirk Q6)
    return swarm_launch_reset(h, h->done_list, h->done_count, h->E);
}

// The step for envs [env_base, env_base + count) only, enqueued on the handle's CURRENT stream: the rollout of the conv policy
// runs every chunk of envs as its own pipeline (forward -> sample -> step -> bookkeeping) on one of a few streams, so the fp64
// VALU-bound step of one chunk overlaps the MFMA GEMMs of the others.  `slot` < 16 selects the done counter (one per stream; the
// finished envs of the range are listed at done_list + env_base).
int swarm_launch_step_range(grl_handle *h, const float *actions_dev, int env_base, int count, int slot, bool counter_zeroed) {
    SwarmParams P = make_params(h);
    P.actions = actions_dev;
    P.actions64 = nullptr;
    P.env_base = env_base;
    P.E = env_base + count;
    P.done_list = h->done_list + env_base;
    P.done_count = h->done_count + slot;
    if (!counter_zeroed) GRL_HIP(h, hipMemsetAsync(P.done_count, 0, sizeof(int32_t), h->stream));      // (the conv rollout's sampling kernel does it)
    if (h->cfg.flags & GRL_F_SWARM_FAST_MATH)
        hipLaunchKernelGGL((swarm_kernel<MODE_STEP, true>), dim3(nblocks(count)), dim3(SWARM_TPB), 0, h->stream, P);
    else
        hipLaunchKernelGGL((swarm_kernel<MODE_STEP, false>), dim3(nblocks(count)), dim3(SWARM_TPB), 0, h->stream, P);
    GRL_HIP(h, hipGetLastError());
    return swarm_launch_reset(h, P.done_list, P.done_count, count);
}

int swarm_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count) {
    SwarmParams P = make_params(h);
    P.reset_list = list_dev;
    P.reset_count = count_dev;
    if (h->cfg.flags & GRL_F_SWARM_FAST_MATH)
        hipLaunchKernelGGL((swarm_kernel<MODE_RESET, true>), dim3(nblocks(max_count)), dim3(SWARM_TPB), 0, h->stream, P);
    else
        hipLaunchKernelGGL((swarm_kernel<MODE_RESET, false>), dim3(nblocks(max_count)), dim3(SWARM_TPB), 0, h->stream, P);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

int swarm_launch_observe(grl_handle *h) {
    SwarmParams P = make_params(h);
    P.flags &= ~GRL_F_SWARM_NO_OBSERVE;
    hipLaunchKernelGGL((swarm_kernel<MODE_OBSERVE, false>), dim3(nblocks(h->E)), dim3(SWARM_TPB), 0, h->stream, P);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

int swarm_reset_injected(grl_handle *h, const double *x0, const double *xa0, const double *ra, const double *an,
                         const double *pn) {
    size_t E = h->E;
    size_t n_x0 = E * N_LOCUSTS * 2, n_xa0 = E * N_AGENTS * 2, n_ra = E * N_BURN_IN * N_AGENTS * 2;
    size_t n_an = E * (N_BURN_IN + 1) * N_AGENTS * 2, n_pn = E * (N_BURN_IN + 1) * N_LOCUSTS * 2;
    double *buf = nullptr;
    GRL_HIP(h, hipMalloc((void **)&buf, (n_x0 + n_xa0 + n_ra + n_an + n_pn) * sizeof(double)));
    double *d_x0 = buf, *d_xa0 = d_x0 + n_x0, *d_ra = d_xa0 + n_xa0, *d_an = d_ra + n_ra, *d_pn = d_an + n_an;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMemcpyAsync(d_x0, x0, n_x0 * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xa0, xa0, n_xa0 * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ra, ra, n_ra * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_an, an, n_an * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pn, pn, n_pn * 8, hipMemcpyHostToDevice, h->stream);
    int rc = GRL_OK;
    if (e == hipSuccess) {
        rc = launch_iota(h, h->done_list, h->E);
        int32_t cnt = h->E;
        if (rc == GRL_OK) e = hipMemcpyAsync(h->done_count, &cnt, sizeof(cnt), hipMemcpyHostToDevice, h->stream);
        if (rc == GRL_OK && e == hipSuccess) {
            e = hipStreamSynchronize(h->stream);   // cnt is a stack variable
            SwarmParams P = make_params(h);
            P.flags &= ~(GRL_F_RESET_FROM_SNAPSHOT);
            P.reset_list = h->done_list; P.reset_count = h->done_count;
            P.inj_x0 = d_x0; P.inj_xa0 = d_xa0; P.inj_ra = d_ra; P.inj_an = d_an; P.inj_pn = d_pn;
            if (h->cfg.flags & GRL_F_SWARM_FAST_MATH)
                hipLaunchKernelGGL((swarm_kernel<MODE_RESET, true>), dim3(nblocks(h->E)), dim3(SWARM_TPB), 0, h->stream, P);
            else
                hipLaunchKernelGGL((swarm_kernel<MODE_RESET, false>), dim3(nblocks(h->E)), dim3(SWARM_TPB), 0, h->stream, P);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        }
    }
    hipFree(buf);
    if (rc != GRL_OK) return rc;
    if (e != hipSuccess) return hip_fail(h, e, "swarm_reset_injected");
    return GRL_OK;
}

int swarm_materialize(grl_handle *h, int first, int count, float *out_dev) {
    int G = h->cfg.grid_size;
    size_t lds = (size_t)2 * G * G * sizeof(unsigned int);
    hipLaunchKernelGGL(swarm_materialize_kernel, dim3(count), dim3(256), lds, h->stream, h->sw.lbins, h->sw.abins,
                       h->sw.pos, first, G, out_dev);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

}  // namespace grl
