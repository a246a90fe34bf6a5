/*
 * oracle/nltgv2_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE float32, no FMA contraction) of the reference's
 * NLTGV2-L1 primal-dual graph regularizer:
 *   /root/reference/src/flame/optimizers/nltgv2_l1_graph_regularizer.{h,cc}
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED: the reference's own tests hold no golden vector / known-answer
 * test for optimizers/ (test/CMakeLists.txt:5-11 never includes it), and the
 * reference translation unit cannot be built in this image: it includes
 * <boost/graph/adjacency_list.hpp> (nltgv2_l1_graph_regularizer.h:27) and, via
 * flame/utils/image_utils.h, OpenCV -- neither Boost nor OpenCV is installed and
 * writing stand-in headers for them is not allowed.  This file is therefore a
 * line-by-line restatement of the published source, and is cross-checked only
 * against a second, independently written numpy restatement (oracle/nltgv2_numpy.py)
 * and against algebraic properties of the algorithm (tests/test_oracle.py).
 *
 * Two layouts are provided:
 *   (1) flat arrays, edge list in the caller's order with explicit (src,dst)
 *       orientation == BGL's boost::edges() order / boost::source / boost::target
 *       (nltgv2...cc:91-96, 118-123).  This is the oracle proper.
 *   (2) "reference layout": individually heap-allocated 52-byte vertex nodes and a
 *       doubly linked edge list, mimicking boost::adjacency_list<hash_setS,hash_setS,
 *       undirectedS,VertexData,EdgeData> (nltgv2...h:107-112).  Same arithmetic,
 *       used only as the single-thread CPU timing stand-in (bench.py cpu_baseline,
 *       kind "port") because the reference runs its solver on one thread
 *       (flame.cc:99-112).
 *
 * Float evaluation order follows the reference text left to right.  Build with
 *   gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math
 * (the reference build is plain x86-64 -std=c++11 without -march, CMakeLists.txt:24,
 * so it has no FMA either).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* nltgv2_l1_graph_regularizer.h:121-129 (defaults .1, .001, 125, .25, 0, 10). */
typedef struct {
  float data_factor;
  float step_x;
  float step_q;
  float theta;
  float x_min;
  float x_max;
} nltgv2_params;

/* Flat view of VertexData (h:74-90) and EdgeData (h:95-102). */
typedef struct {
  int32_t V, E;
  const float* pos; /* 2*V, interleaved (x,y): cv::Point2f pos, h:75 */
  float *x, *w1, *w2;
  float *x_bar, *w1_bar, *w2_bar;
  float *x_prev, *w1_prev, *w2_prev;
  const float *data_term, *data_weight;
  const int32_t *src, *dst; /* boost::source / boost::target of edge k */
  const float *alpha, *beta;
  float *q1, *q2, *q3;
} nltgv2_graph;

/* image_utils.h:93-96 */
static inline float fast_abs(float r) { return (r > 0) ? r : -r; }

/* h:171-176.  `step` is ignored by the reference as well.  Returns NaN status via *bad
 * instead of FLAME_ASSERT -> exit(1) (assert.h:111). */
static inline float prox_nltgv2_conj(float step, float q, int* bad) {
  (void)step;
  float absq = fast_abs(q);
  float new_q = q / (absq > 1 ? absq : 1);
  if (isnan(new_q)) *bad = 1;
  return new_q;
}

/* h:179-197 */
static inline float prox_l1(float x_min, float x_max, float step_x, float data_weight, float x,
                            float data) {
  float diff = x - data;
  float thresh = step_x * data_weight;
  float new_x = 0.0f;
  if (diff > thresh) {
    new_x = x - thresh;
  } else if (diff < -thresh) {
    new_x = x + thresh;
  } else {
    new_x = data;
  }
  new_x = (new_x < x_min) ? x_min : new_x;
  new_x = (new_x > x_max) ? x_max : new_x;
  return new_x;
}

/* cc:89-114 */
int nltgv2_oracle_dual_step(const nltgv2_params* p, nltgv2_graph* g) {
  int bad = 0;
  for (int32_t k = 0; k < g->E; ++k) {
    const int32_t ii = g->src[k], jj = g->dst[k];
    const float alpha = g->alpha[k], beta = g->beta[k];
    float K1x = alpha * (g->x_bar[ii] - g->x_bar[jj]);
    K1x -= alpha * (g->pos[2 * ii] - g->pos[2 * jj]) * g->w1_bar[ii];
    K1x -= alpha * (g->pos[2 * ii + 1] - g->pos[2 * jj + 1]) * g->w2_bar[ii];
    g->q1[k] = prox_nltgv2_conj(p->step_q, g->q1[k] + p->step_q * K1x, &bad);

    float K2x = beta * (g->w1_bar[ii] - g->w1_bar[jj]);
    g->q2[k] = prox_nltgv2_conj(p->step_q, g->q2[k] + p->step_q * K2x, &bad);

    float K3x = beta * (g->w2_bar[ii] - g->w2_bar[jj]);
    g->q3[k] = prox_nltgv2_conj(p->step_q, g->q3[k] + p->step_q * K3x, &bad);
  }
  return bad;
}

/* cc:116-154 */
void nltgv2_oracle_primal_step(const nltgv2_params* p, nltgv2_graph* g) {
  for (int32_t k = 0; k < g->E; ++k) {
    const int32_t ii = g->src[k], jj = g->dst[k];
    const float alpha = g->alpha[k], beta = g->beta[k];
    const float q1 = g->q1[k], q2 = g->q2[k], q3 = g->q3[k];

    g->x[ii] -= q1 * p->step_x * alpha;
    g->x[jj] += q1 * p->step_x * alpha;

    g->w1[ii] += q1 * p->step_x * alpha * (g->pos[2 * ii] - g->pos[2 * jj]);
    g->w2[ii] += q1 * p->step_x * alpha * (g->pos[2 * ii + 1] - g->pos[2 * jj + 1]);

    g->w1[ii] -= q2 * p->step_x * beta;
    g->w1[jj] += q2 * p->step_x * beta;

    g->w2[ii] -= q3 * p->step_x * beta;
    g->w2[jj] += q3 * p->step_x * beta;
  }
  for (int32_t v = 0; v < g->V; ++v) {
    g->x[v] = prox_l1(p->x_min, p->x_max, p->step_x, p->data_factor * g->data_weight[v], g->x[v],
                      g->data_term[v]);
  }
}

/* cc:156-174 */
void nltgv2_oracle_extragradient_step(const nltgv2_params* p, nltgv2_graph* g) {
  for (int32_t v = 0; v < g->V; ++v) {
    float new_x_bar = g->x[v] + p->theta * (g->x[v] - g->x_prev[v]);
    new_x_bar = (new_x_bar < p->x_min) ? p->x_min : new_x_bar;
    new_x_bar = (new_x_bar > p->x_max) ? p->x_max : new_x_bar;
    g->x_bar[v] = new_x_bar;
    g->w1_bar[v] = g->w1[v] + p->theta * (g->w1[v] - g->w1_prev[v]);
    g->w2_bar[v] = g->w2[v] + p->theta * (g->w2[v] - g->w2_prev[v]);
  }
}

/* cc:33-49 */
int nltgv2_oracle_step(const nltgv2_params* p, nltgv2_graph* g) {
  for (int32_t v = 0; v < g->V; ++v) {
    g->x_prev[v] = g->x[v];
    g->w1_prev[v] = g->w1[v];
    g->w2_prev[v] = g->w2[v];
  }
  int bad = nltgv2_oracle_dual_step(p, g);
  nltgv2_oracle_primal_step(p, g);
  nltgv2_oracle_extragradient_step(p, g);
  return bad;
}

int nltgv2_oracle_run(const nltgv2_params* p, nltgv2_graph* g, int n_iters) {
  int bad = 0;
  for (int it = 0; it < n_iters; ++it) bad |= nltgv2_oracle_step(p, g);
  return bad;
}

/* cc:51-71; sequential float accumulation in edge order. */
float nltgv2_oracle_smoothness_cost(const nltgv2_params* p, const nltgv2_graph* g) {
  float cost = 0.0f;
  for (int32_t k = 0; k < g->E; ++k) {
    const int32_t ii = g->src[k], jj = g->dst[k];
    const float dx = g->pos[2 * ii] - g->pos[2 * jj];
    const float dy = g->pos[2 * ii + 1] - g->pos[2 * jj + 1];
    float a = g->x[ii] - g->x[jj] - g->w1[ii] * dx - g->w2[ii] * dy;
    a = (a >= 0) ? a : -a;
    cost += g->alpha[k] * a;
    float b = g->w1[ii] - g->w1[jj];
    b = (b >= 0) ? b : -b;
    float c = g->w2[ii] - g->w2[jj];
    c = (c >= 0) ? c : -c;
    cost += g->beta[k] * b + g->beta[k] * c;
  }
  return p->data_factor * cost;
}

/* cc:73-85; sequential float accumulation in vertex order. */
float nltgv2_oracle_data_cost(const nltgv2_params* p, const nltgv2_graph* g) {
  (void)p;
  float cost = 0.0f;
  for (int32_t v = 0; v < g->V; ++v) {
    float diff = (g->x[v] - g->data_term[v]) * g->data_weight[v];
    diff = (diff > 0) ? diff : -diff;
    cost += diff;
  }
  return cost;
}

/* ------------------------------------------------------------------------------------------
 * (2) reference-layout variant: timing stand-in only.
 * ------------------------------------------------------------------------------------------ */
typedef struct rl_vertex { /* VertexData, h:74-90: 13 floats = 52 B */
  float pos_x, pos_y;
  float x, w1, w2;
  float x_bar, w1_bar, w2_bar;
  float x_prev, w1_prev, w2_prev;
  float data_term, data_weight;
} rl_vertex;

typedef struct rl_edge { /* std::list node: prev/next + (source,target) + EdgeData h:95-102 */
  struct rl_edge *prev, *next;
  rl_vertex *source, *target;
  float alpha, beta, q1, q2, q3;
  int valid;
} rl_edge;

typedef struct {
  int32_t V, E;
  rl_vertex** vtx;    /* handles in original index order */
  rl_vertex** vorder; /* iteration order of boost::vertices(): a hash-set walk */
  rl_edge* head;
} rl_graph;

static uint64_t rl_mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

rl_graph* nltgv2_reflayout_create(const nltgv2_graph* g) {
  rl_graph* r = (rl_graph*)calloc(1, sizeof(rl_graph));
  r->V = g->V;
  r->E = g->E;
  r->vtx = (rl_vertex**)malloc(sizeof(rl_vertex*) * (size_t)(g->V > 0 ? g->V : 1));
  r->vorder = (rl_vertex**)malloc(sizeof(rl_vertex*) * (size_t)(g->V > 0 ? g->V : 1));
  for (int32_t v = 0; v < g->V; ++v) {
    rl_vertex* n = (rl_vertex*)malloc(sizeof(rl_vertex));
    n->pos_x = g->pos[2 * v];
    n->pos_y = g->pos[2 * v + 1];
    n->x = g->x[v], n->w1 = g->w1[v], n->w2 = g->w2[v];
    n->x_bar = g->x_bar[v], n->w1_bar = g->w1_bar[v], n->w2_bar = g->w2_bar[v];
    n->x_prev = g->x_prev[v], n->w1_prev = g->w1_prev[v], n->w2_prev = g->w2_prev[v];
    n->data_term = g->data_term[v], n->data_weight = g->data_weight[v];
    r->vtx[v] = n;
    r->vorder[v] = n;
  }
  /* hash-set iteration order: a fixed pseudo-random permutation (Fisher-Yates on splitmix). */
  for (int32_t v = g->V - 1; v > 0; --v) {
    int32_t j = (int32_t)(rl_mix((uint64_t)v) % (uint64_t)(v + 1));
    rl_vertex* t = r->vorder[v];
    r->vorder[v] = r->vorder[j];
    r->vorder[j] = t;
  }
  rl_edge* tail = NULL;
  for (int32_t k = 0; k < g->E; ++k) {
    rl_edge* e = (rl_edge*)malloc(sizeof(rl_edge));
    e->prev = tail, e->next = NULL;
    e->source = r->vtx[g->src[k]], e->target = r->vtx[g->dst[k]];
    e->alpha = g->alpha[k], e->beta = g->beta[k];
    e->q1 = g->q1[k], e->q2 = g->q2[k], e->q3 = g->q3[k];
    e->valid = 1;
    if (tail) tail->next = e; else r->head = e;
    tail = e;
  }
  return r;
}

void nltgv2_reflayout_destroy(rl_graph* r) {
  if (!r) return;
  for (rl_edge* e = r->head; e;) {
    rl_edge* n = e->next;
    free(e);
    e = n;
  }
  for (int32_t v = 0; v < r->V; ++v) free(r->vtx[v]);
  free(r->vtx);
  free(r->vorder);
  free(r);
}

int nltgv2_reflayout_step(const nltgv2_params* p, rl_graph* r) {
  int bad = 0;
  for (int32_t v = 0; v < r->V; ++v) { /* cc:35-42 */
    rl_vertex* n = r->vorder[v];
    n->x_prev = n->x, n->w1_prev = n->w1, n->w2_prev = n->w2;
  }
  for (rl_edge* e = r->head; e; e = e->next) { /* cc:93-111 */
    rl_vertex *a = e->source, *b = e->target;
    float K1x = e->alpha * (a->x_bar - b->x_bar);
    K1x -= e->alpha * (a->pos_x - b->pos_x) * a->w1_bar;
    K1x -= e->alpha * (a->pos_y - b->pos_y) * a->w2_bar;
    e->q1 = prox_nltgv2_conj(p->step_q, e->q1 + p->step_q * K1x, &bad);
    float K2x = e->beta * (a->w1_bar - b->w1_bar);
    e->q2 = prox_nltgv2_conj(p->step_q, e->q2 + p->step_q * K2x, &bad);
    float K3x = e->beta * (a->w2_bar - b->w2_bar);
    e->q3 = prox_nltgv2_conj(p->step_q, e->q3 + p->step_q * K3x, &bad);
  }
  for (rl_edge* e = r->head; e; e = e->next) { /* cc:120-142 */
    rl_vertex *a = e->source, *b = e->target;
    a->x -= e->q1 * p->step_x * e->alpha;
    b->x += e->q1 * p->step_x * e->alpha;
    a->w1 += e->q1 * p->step_x * e->alpha * (a->pos_x - b->pos_x);
    a->w2 += e->q1 * p->step_x * e->alpha * (a->pos_y - b->pos_y);
    a->w1 -= e->q2 * p->step_x * e->beta;
    b->w1 += e->q2 * p->step_x * e->beta;
    a->w2 -= e->q3 * p->step_x * e->beta;
    b->w2 += e->q3 * p->step_x * e->beta;
  }
  for (int32_t v = 0; v < r->V; ++v) { /* cc:147-151 */
    rl_vertex* n = r->vorder[v];
    n->x = prox_l1(p->x_min, p->x_max, p->step_x, p->data_factor * n->data_weight, n->x,
                   n->data_term);
  }
  for (int32_t v = 0; v < r->V; ++v) { /* cc:160-171 */
    rl_vertex* n = r->vorder[v];
    float nb = n->x + p->theta * (n->x - n->x_prev);
    nb = (nb < p->x_min) ? p->x_min : nb;
    nb = (nb > p->x_max) ? p->x_max : nb;
    n->x_bar = nb;
    n->w1_bar = n->w1 + p->theta * (n->w1 - n->w1_prev);
    n->w2_bar = n->w2 + p->theta * (n->w2 - n->w2_prev);
  }
  return bad;
}

/* Runs n_iters steps, returns elapsed seconds (CLOCK_MONOTONIC). */
double nltgv2_reflayout_run_timed(const nltgv2_params* p, rl_graph* r, int n_iters) {
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int it = 0; it < n_iters; ++it) nltgv2_reflayout_step(p, r);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* Copies x,w1,w2,x_bar.. and q back to the flat graph (for cross-checking layout (2) vs (1)). */
void nltgv2_reflayout_export(const rl_graph* r, nltgv2_graph* g) {
  for (int32_t v = 0; v < r->V; ++v) {
    const rl_vertex* n = r->vtx[v];
    g->x[v] = n->x, g->w1[v] = n->w1, g->w2[v] = n->w2;
    g->x_bar[v] = n->x_bar, g->w1_bar[v] = n->w1_bar, g->w2_bar[v] = n->w2_bar;
    g->x_prev[v] = n->x_prev, g->w1_prev[v] = n->w1_prev, g->w2_prev[v] = n->w2_prev;
  }
  int32_t k = 0;
  for (const rl_edge* e = r->head; e; e = e->next, ++k) {
    g->q1[k] = e->q1, g->q2[k] = e->q2, g->q3[k] = e->q3;
  }
}

double nltgv2_oracle_run_timed(const nltgv2_params* p, nltgv2_graph* g, int n_iters) {
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  nltgv2_oracle_run(p, g, n_iters);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
