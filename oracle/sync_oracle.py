"""Restatement (test infrastructure) of the graph EDITS of Flame::syncGraph / projectGraph
(/root/reference/src/flame/flame.cc:1985-2121, 1923-1931, 2123-2163) on containers that mimic what the
reference's Boost.Graph container guarantees:
  * edges live in a std::list: boost::edges() = insertion order, erase keeps the order of the rest,
    add_edge appends, boost::edge(u,v) finds an undirected edge whichever way it was added and the found
    edge keeps its original (source,target);
  * vertices are looked up by feature id (feat_to_vtx_); their iteration order in the reference is BGL
    hash order, i.e. unspecified -- the flat image here uses the caller's feature order.
PARITY UNPINNED like the solver checker: the reference has no test for syncGraph and cannot be built here.
"""
from __future__ import annotations

import numpy as np

F = np.float32
VKEYS = ("x", "w1", "w2", "x_bar", "w1_bar", "w2_bar", "x_prev", "w1_prev", "w2_prev")


class RefGraph:
    def __init__(self):
        self.v = {}      # feat id -> dict(pos, data_term, data_weight, x, w1, ...)
        self.e = []      # [fa, fb, dict(alpha,beta,q1,q2,q3,valid)]   (fa = boost::source, fb = boost::target)

    @staticmethod
    def from_flat(g: dict, feat_id=None):
        r = RefGraph()
        fid = np.arange(g["V"]) if feat_id is None else feat_id
        for i in range(g["V"]):
            d = dict(pos=g["pos"][i].copy(), data_term=F(g["data_term"][i]), data_weight=F(g["data_weight"][i]))
            for k in VKEYS:
                d[k] = F(g[k][i])
            r.v[int(fid[i])] = d
        for k in range(g["E"]):
            r.e.append([int(fid[g["src"][k]]), int(fid[g["dst"][k]]),
                        dict(alpha=F(g["alpha"][k]), beta=F(g["beta"][k]), q1=F(g["q1"][k]), q2=F(g["q2"][k]),
                             q3=F(g["q3"][k]), valid=True)])
        return r

    def find_edge(self, a, b):
        for ed in self.e:  # tests use small graphs; an index is built in sync() for speed
            if (ed[0] == a and ed[1] == b) or (ed[0] == b and ed[1] == a):
                return ed
        return None


def sync(gr: RefGraph, feat_id, pos, data_term, data_weight, tri_edges, init_x=None, check_sticky=False, thr=0.25,
         init_graph_scale=0.0):
    feat_id = [int(f) for f in feat_id]
    idx = {f: i for i, f in enumerate(feat_id)}
    feats_to_update = set(feat_id)
    # ---- update existing vertices / mark for removal (flame.cc:1985-2018)
    remove = []
    for f, d in gr.v.items():
        if f not in feats_to_update:
            remove.append(f)
            continue
        i = idx[f]
        d["pos"] = np.asarray(pos[i], F).copy()
        d["data_term"] = F(data_term[i])
        d["data_weight"] = F(data_weight[i])
        if check_sticky and (F(d["x"]) - F(d["data_term"]) > F(thr)):
            d["x"] = F(d["data_term"])
        feats_to_update.discard(f)
    # ---- remove marked vertices with their edges (flame.cc:2020-2028; projectGraph 1923-1931)
    rm = set(remove)
    for f in remove:
        del gr.v[f]
    gr.e = [ed for ed in gr.e if ed[0] not in rm and ed[1] not in rm]
    # ---- add new vertices (flame.cc:2030-2049, VertexData defaults h:74-90)
    new = [f for f in feat_id if f in feats_to_update]
    for f in new:
        i = idx[f]
        d = dict(pos=np.asarray(pos[i], F).copy(), data_term=F(data_term[i]), data_weight=F(data_weight[i]))
        for k in VKEYS:
            d[k] = F(0)
        d["x"] = d["x_bar"] = d["x_prev"] = F(data_term[i])
        gr.v[f] = d
    # ---- edges (flame.cc:2075-2121)
    for ed in gr.e:
        ed[2]["valid"] = False
    index = {}
    for ed in gr.e:
        index[(min(ed[0], ed[1]), max(ed[0], ed[1]))] = ed
    for a, b in np.asarray(tri_edges).reshape(-1, 2):
        fa, fb = feat_id[int(a)], feat_id[int(b)]
        key = (min(fa, fb), max(fa, fb))
        ed = index.get(key)
        if ed is None:
            ed = [fa, fb, dict(alpha=F(1), beta=F(1), q1=F(0), q2=F(0), q3=F(0), valid=True)]
            gr.e.append(ed)  # boost::add_edge(vtx_ii, vtx_jj, EdgeData(), graph): appended
            index[key] = ed
        d = np.asarray(pos[int(a)], F) - np.asarray(pos[int(b)], F)
        length = np.sqrt(d[0] * d[0] + d[1] * d[1], dtype=F)
        ed[2]["alpha"] = F(F(1.0) / length)
        ed[2]["beta"] = F(1.0)
        ed[2]["valid"] = True
    gr.e = [ed for ed in gr.e if ed[2]["valid"]]
    # ---- initial x of the new vertices (flame.cc:2123-2163).  init_x = the caller's idepthmap lookup / graph_scale.
    # With init_graph_scale > 0 a NaN entry falls back to the mean of the neighbours (flame.cc:2133-2158).  The reference
    # walks feats_to_update (an unordered_set) and each vertex's adjacency (hash-set out edges) in unspecified order, so
    # its result depends on the hash order when new vertices neighbour each other.  The order fixed here (and in
    # k_sync_init_from_neighbours): first every vertex with a valid prediction takes it; then all the others form their
    # means at once, from the values standing at that point (a new vertex without prediction stands at its data term,
    # flame.cc:2046-2048), neighbours in ascending edge id.
    need = []
    for f in new:
        i = idx[f]
        xi = F(data_term[i]) if init_x is None else F(init_x[i])
        if init_graph_scale > 0 and np.isnan(xi):
            need.append(f)
            xi = F(data_term[i])
        gr.v[f]["x"] = gr.v[f]["x_bar"] = gr.v[f]["x_prev"] = xi
    if need:
        gs = F(init_graph_scale)
        adj = {f: [] for f in need}
        for ed in gr.e:  # ascending edge id
            if ed[0] in adj:
                adj[ed[0]].append(ed[1])
            if ed[1] in adj:
                adj[ed[1]].append(ed[0])
        res = {}
        for f in need:
            s, cnt = F(0), 0
            for nb in adj[f]:
                if gr.v[nb]["data_weight"] > 0:
                    s = F(s + F(gr.v[nb]["x"] * gs))
                    cnt += 1
            res[f] = F(F(s / F(cnt)) / gs) if cnt > 0 else F(gr.v[f]["data_term"])
        for f in need:
            gr.v[f]["x"] = gr.v[f]["x_bar"] = gr.v[f]["x_prev"] = res[f]
    return gr


def flatten(gr: RefGraph, feat_order) -> dict:
    feat_order = [int(f) for f in feat_order]
    pos_of = {f: i for i, f in enumerate(feat_order)}
    V, E = len(feat_order), len(gr.e)
    g = dict(V=V, E=E, pos=np.zeros((V, 2), F), data_term=np.zeros(V, F), data_weight=np.zeros(V, F))
    for k in VKEYS:
        g[k] = np.zeros(V, F)
    for f, i in pos_of.items():
        d = gr.v[f]
        g["pos"][i] = d["pos"]
        g["data_term"][i], g["data_weight"][i] = d["data_term"], d["data_weight"]
        for k in VKEYS:
            g[k][i] = d[k]
    g["src"] = np.array([pos_of[ed[0]] for ed in gr.e], np.int32)
    g["dst"] = np.array([pos_of[ed[1]] for ed in gr.e], np.int32)
    for k in ("alpha", "beta", "q1", "q2", "q3"):
        g[k] = np.array([ed[2][k] for ed in gr.e], F)
    return g


def absorb(gr: RefGraph, g: dict, feat_order):
    """Writes solver results (flat arrays in feat_order / current edge order) back into the containers."""
    for i, f in enumerate(feat_order):
        for k in VKEYS:
            gr.v[int(f)][k] = F(g[k][i])
    for k, ed in enumerate(gr.e):
        for q in ("q1", "q2", "q3"):
            ed[2][q] = F(g[q][k])
