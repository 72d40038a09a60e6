/* CPU ORACLE (C) -- TEST INFRASTRUCTURE ONLY, NOT PART OF THE PRODUCT PATH.
 *
 * Scalar float64 restatement of the reference's Swarm hot path, one env at a time, following
 * /root/reference/fed_gym/envs/multiagent.py:30-115 (SwarmEnv._step, x_update, xv_cutoff,
 * v_calculate, s) and /root/reference/fed_gym/agents/state_processors.py:17-42 (process_state),
 * plus the n-step return loop of /root/reference/fed_gym/agents/paac/paac.py:159-172,360-365.
 * Used by tests/ (checked against oracle/oracle.py, which is itself pinned bit-exact on the
 * golden vectors) and by bench.py's `cpu_baseline` leg (kind "port").  Summation orders follow
 * numpy (pairwise with 8 accumulators for contiguous 1-D sums, sequential for axis-0 means).
 * Build: make -C oracle  ->  oracle/liboracle.so  (gcc -O2, no -ffast-math, -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NL 80
#define NA 10
static const double DT = 0.05, NOISE = 0.0001, WIND = 1.0, GRAV = -1.0, FATT = 0.5, LATT = 10.0;

static void cutoff(double *y, double *vx, double *vy) { /* multiagent.py:77-86 */
    if (*y <= 0) { *y = 0; *vx = 0; if (*vy <= 0) *vy = 0; }
}

static void x_update(double *p, double vx, double vy, double nx, double ny) { /* multiagent.py:70-75 */
    cutoff(&p[1], &vx, &vy);
    p[0] = p[0] + (DT * vx + NOISE * nx);
    p[1] = p[1] + (DT * vy + NOISE * ny);
    if (p[1] <= 0) p[1] = 0;
}

static double pairwise(const double *a, int n) { /* numpy pairwise_sum for 8 <= n <= 128 */
    double r[8];
    int i, k;
    for (k = 0; k < 8; ++k) r[k] = a[k];
    for (i = 8; i < n - (n % 8); i += 8)
        for (k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

static double s_fn(double r) { return FATT * exp(-r / LATT) - exp(-r); } /* multiagent.py:65-68 */

/* x_update for the agents with the worker's FLOAT32 action row (quirk Q7): `v_action[:, 0] += 1` and `dt * v` stay float32 in
 * numpy (multiagent.py:33-36, 72); pinned by tests/golden/swarm_runner.npz through oracle.py */
static void x_update_f32v(double *p, float vx, float vy, double nx, double ny) {
    if (p[1] <= 0) {
        p[1] = 0;
        vx = 0;
        if (vy <= 0) vy = 0;
    }
    p[0] = p[0] + ((double)(0.05f * vx) + NOISE * nx);
    p[1] = p[1] + ((double)(0.05f * vy) + NOISE * ny);
    if (p[1] <= 0) p[1] = 0;
}

/* one SwarmEnv._step: x (80,2) xa (10,2) updated in place; action (10,2) f32 as the worker reads it; noise rows raw N(0,1) */
static double swarm_step_one(double *x, double *xa, const float *action, const double *an, const double *pn) {
    double v[NL][2], t0[NL], t1[NL], en[NL];
    int i, j, a;
    for (a = 0; a < NA; ++a) x_update_f32v(&xa[2 * a], action[2 * a] + 1.0f, action[2 * a + 1], an[2 * a], an[2 * a + 1]);
    for (j = 0; j < NL; ++j) { /* v_calculate, multiagent.py:88-113 */
        double xj = x[2 * j], yj = x[2 * j + 1];
        for (i = 0; i < NL; ++i) {
            double dx = x[2 * i] - xj, dy = x[2 * i + 1] - yj;
            double d = sqrt(dx * dx + dy * dy), s = s_fn(d);
            t0[i] = s * dx / (d + 0.000001);
            t1[i] = s * dy / (d + 0.000001);
        }
        double ll0 = pairwise(t0, NL), ll1 = pairwise(t1, NL);
        for (a = 0; a < NA; ++a) {
            double dx = xa[2 * a] - xj, dy = xa[2 * a + 1] - yj;
            double d = sqrt(dx * dx + dy * dy), s = s_fn(d);
            t0[a] = s * dx / (d + 0.000001);
            t1[a] = s * dy / (d + 0.000001);
        }
        v[j][0] = (WIND + ll0) + pairwise(t0, NA);
        v[j][1] = (GRAV + ll1) + pairwise(t1, NA);
        en[j] = v[j][0] * v[j][0] + v[j][1] * v[j][1];
    }
    double energy = pairwise(en, NL) / (double)NL; /* multiagent.py:114 */
    for (j = 0; j < NL; ++j) x_update(&x[2 * j], v[j][0], v[j][1], pn[2 * j], pn[2 * j + 1]);
    return -energy;
}

static int count_le(double v, const double *edges, int n) { /* searchsorted(edges, v, 'right') */
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (edges[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* process_state for one env: bins as uint8 pairs (255 = outside), positions = digitize clamped */
static void swarm_observe_one(const double *x, const double *xa, int G, uint8_t *lb, uint8_t *ab, uint8_t *pos) {
    double ex[256], ey[256];
    double sum = x[0];
    int i, k;
    for (i = 1; i < NL; ++i) sum += x[2 * i];       /* np.mean(vstack, axis=0)[0]: sequential */
    for (i = 0; i < NA; ++i) sum += xa[2 * i];
    double m = sum / 90.0, lo = m - 1.5, hi = m + 1.5, step = (hi - lo) / (double)G, ystep = (6.0 - 0.0) / (double)G;
    for (k = 0; k <= G; ++k) { ex[k] = (double)k * step + lo; ey[k] = (double)k * ystep + 0.0; }
    ex[G] = hi; ey[G] = 6.0;                        /* linspace endpoint */
    for (i = 0; i < NL + NA; ++i) {
        const double *p = i < NL ? &x[2 * i] : &xa[2 * (i - NL)];
        int cx = count_le(p[0], ex, G + 1), cy = count_le(p[1], ey, G + 1);
        int bx = p[0] == ex[G] ? G - 1 : cx - 1, by = p[1] == ey[G] ? G - 1 : cy - 1;
        int in = bx >= 0 && bx < G && by >= 0 && by < G;
        uint8_t *o = i < NL ? &lb[2 * i] : &ab[2 * (i - NL)];
        o[0] = in ? (uint8_t)bx : 255; o[1] = in ? (uint8_t)by : 255;
        if (i >= NL) {
            pos[2 * (i - NL)] = (uint8_t)(cx >= G ? G - 1 : cx);
            pos[2 * (i - NL) + 1] = (uint8_t)(cy >= G ? G - 1 : cy);
        }
    }
}

/* batch entry points; threads = 0 -> all cores (OpenMP), 1 -> serial.  Returns threads used. */
int oracle_swarm_step(int E, double *x, double *xa, const float *action, const double *an, const double *pn,
                      double *reward, int G, uint8_t *lb, uint8_t *ab, uint8_t *pos, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel for schedule(static)
#endif
    for (int e = 0; e < E; ++e) {
        const float *act = action + (long)e * 2 * NA;
        reward[e] = swarm_step_one(x + (long)e * 2 * NL, xa + (long)e * 2 * NA, act, an + (long)e * 2 * NA, pn + (long)e * 2 * NL);
        if (lb) swarm_observe_one(x + (long)e * 2 * NL, xa + (long)e * 2 * NA, G, lb + (long)e * 2 * NL, ab + (long)e * 2 * NA, pos + (long)e * 2 * NA);
    }
    return used;
}

/* paac.py:167-172 / 360-365: est <- r + gamma*est*mask (first product in float32: boot is the net's f32) */
void oracle_returns(int T, int B, const float *r, const float *v, const float *mask, const float *boot, double gamma,
                    double scale, double *y, double *adv) {
    for (int b = 0; b < B; ++b) {
        double est = 0;
        for (int t = T - 1; t >= 0; --t) {
            long i = (long)t * B + b;
            double ge = t == T - 1 ? (double)((float)gamma * boot[b]) : gamma * est;
            if (mask) ge *= (double)mask[i];
            est = (double)r[i] + ge;
            y[i] = est;
            adv[i] = (est - (double)v[i]) / scale;
        }
    }
}
