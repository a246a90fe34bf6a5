/*
 * oracle/nltgv2_omp.c -- TEST INFRASTRUCTURE: the "strongest fair CPU number" of BASELINE.md section 3(3), an
 * OpenMP two-phase form of the reference's step() (nltgv2_l1_graph_regularizer.cc:33-49) over all host cores.
 *
 *   phase 1, parallel over edges:    dualStep (cc:89-114) -- edges are independent
 *   phase 2, parallel over vertices: the prev copy (cc:35-42), the primal scatter (cc:120-142) re-expressed as a
 *            per-vertex gather over the vertex's incident edges in ASCENDING EDGE ID (every variable of a vertex
 *            then sees the same operations in the same order as in the reference's sequential scatter), proxL1
 *            (cc:147-151) and the extragradient step (cc:160-171)
 *
 * Bit-identical to the sequential restatement nltgv2_oracle.c for any thread count (tests/test_oracle.py).  The
 * reference itself runs this solver on ONE thread (flame.cc:99-112; its omp pragmas are inert, CMakeLists.txt:24);
 * this file exists only so that bench.py can print a multi-core CPU figure next to the GPU's.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct nltgv2_params {
  float data_factor, step_x, step_q, theta, x_min, x_max;
} nltgv2_params;

typedef struct nltgv2_graph {
  int32_t V, E;
  float* pos;
  float *x, *w1, *w2, *x_bar, *w1_bar, *w2_bar, *x_prev, *w1_prev, *w2_prev, *data_term, *data_weight;
  int32_t *src, *dst;
  float *alpha, *beta, *q1, *q2, *q3;
} nltgv2_graph;

static inline float conj_prox(float q, int* bad) {
  const float aq = (q > 0) ? q : -q;
  const float d = (aq > 1.0f) ? aq : 1.0f;
  const float r = q / d;
  if (isnan(r)) *bad = 1;
  return r;
}

static inline float prox_l1(float x_min, float x_max, float sigma, float lambda, float x, float data) {
  const float thresh = sigma * lambda;
  const float diff = x - data;
  float nx;
  if (diff > thresh) nx = x - thresh;
  else if (diff < -thresh) nx = x + thresh;
  else nx = data;
  nx = (nx < x_min) ? x_min : nx;
  nx = (nx > x_max) ? x_max : nx;
  return nx;
}

/* Returns 0, 1 if a dual became NaN, -1 on allocation failure. */
int nltgv2_omp_run(const nltgv2_params* p, nltgv2_graph* g, int n_iters, int n_threads) {
  const int32_t V = g->V, E = g->E;
  int32_t* row = (int32_t*)calloc((size_t)V + 1, sizeof(int32_t));
  int32_t* inc = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(E > 0 ? E : 1));
  int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));
  if (!row || !inc || !fill) {
    free(row), free(inc), free(fill);
    return -1;
  }
  for (int32_t k = 0; k < E; ++k) row[g->src[k] + 1]++, row[g->dst[k] + 1]++;
  for (int32_t v = 0; v < V; ++v) row[v + 1] += row[v], fill[v] = row[v];
  /* ascending edge id per vertex; bit 31 set: the vertex is the edge's TARGET */
  for (int32_t k = 0; k < E; ++k) {
    inc[fill[g->src[k]]++] = k;
    inc[fill[g->dst[k]]++] = (int32_t)((uint32_t)k | 0x80000000u);
  }
  int bad = 0;
#pragma omp parallel num_threads(n_threads) reduction(| : bad)
  {
    for (int it = 0; it < n_iters; ++it) {
#pragma omp for schedule(static)
      for (int32_t k = 0; k < E; ++k) {
        const int32_t ii = g->src[k], jj = g->dst[k];
        const float alpha = g->alpha[k], beta = g->beta[k];
        float K1x = alpha * (g->x_bar[ii] - g->x_bar[jj]);
        K1x -= alpha * (g->pos[2 * ii] - g->pos[2 * jj]) * g->w1_bar[ii];
        K1x -= alpha * (g->pos[2 * ii + 1] - g->pos[2 * jj + 1]) * g->w2_bar[ii];
        g->q1[k] = conj_prox(g->q1[k] + p->step_q * K1x, &bad);
        const float K2x = beta * (g->w1_bar[ii] - g->w1_bar[jj]);
        g->q2[k] = conj_prox(g->q2[k] + p->step_q * K2x, &bad);
        const float K3x = beta * (g->w2_bar[ii] - g->w2_bar[jj]);
        g->q3[k] = conj_prox(g->q3[k] + p->step_q * K3x, &bad);
      }
#pragma omp for schedule(static)
      for (int32_t v = 0; v < V; ++v) {
        const float x0 = g->x[v], w10 = g->w1[v], w20 = g->w2[v];
        g->x_prev[v] = x0, g->w1_prev[v] = w10, g->w2_prev[v] = w20;
        float x = x0, w1 = w10, w2 = w20;
        for (int32_t h = row[v]; h < row[v + 1]; ++h) {
          const int32_t k = (int32_t)((uint32_t)inc[h] & 0x7fffffffu);
          const float alpha = g->alpha[k], beta = g->beta[k];
          const float q1 = g->q1[k], q2 = g->q2[k], q3 = g->q3[k];
          if (inc[h] < 0) { /* v == jj */
            x += q1 * p->step_x * alpha;
            w1 += q2 * p->step_x * beta;
            w2 += q3 * p->step_x * beta;
          } else { /* v == ii */
            const int32_t jj = g->dst[k];
            x -= q1 * p->step_x * alpha;
            w1 += q1 * p->step_x * alpha * (g->pos[2 * v] - g->pos[2 * jj]);
            w2 += q1 * p->step_x * alpha * (g->pos[2 * v + 1] - g->pos[2 * jj + 1]);
            w1 -= q2 * p->step_x * beta;
            w2 -= q3 * p->step_x * beta;
          }
        }
        x = prox_l1(p->x_min, p->x_max, p->step_x, p->data_factor * g->data_weight[v], x, g->data_term[v]);
        g->x[v] = x, g->w1[v] = w1, g->w2[v] = w2;
        float nb = x + p->theta * (x - x0);
        nb = (nb < p->x_min) ? p->x_min : nb;
        nb = (nb > p->x_max) ? p->x_max : nb;
        g->x_bar[v] = nb;
        g->w1_bar[v] = w1 + p->theta * (w1 - w10);
        g->w2_bar[v] = w2 + p->theta * (w2 - w20);
      }
    }
  }
  free(row), free(inc), free(fill);
  return bad;
}

int nltgv2_omp_max_threads(void) { return omp_get_max_threads(); }
