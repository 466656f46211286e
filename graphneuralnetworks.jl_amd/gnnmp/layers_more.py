"""More layers of the reference's zoo that run on the same kernels: CGConv, EdgeConv, GatedGraphConv, DConv, NNConv, MEGNetConv, GMMConv, EGNNConv, ChebConv, and the Set2Set pool.
GNNlib/src/layers/conv.jl  cg_conv :304-333, edge_conv :237-246, gated_graph_conv :218-233, d_conv :696-725;
constructors GraphNeuralNetworks/src/layers/conv.jl :925-931 (CGConv), :582 (EdgeConv), :525-530 (GatedGraphConv),
:1584-1589 (DConv).

  * CGConv: `dense_f(z) .* dense_s(z)` with z = vcat(xi, xj, e) is split by the column blocks of the two weight matrices
    into node-level contractions (x_i share, x_j share) and one edge-level contraction (e share); the per-edge sum,
    sigmoid, activation, product and the aggregation are ONE pass of the row kernel (gnnmp_propagate_cg_f32).  The
    (2 nin + ein, E) concatenation and the two (out, E) dense outputs of the reference are never built.
  * EdgeConv: a general `nn` sees per-edge inputs, so the reference's composition is kept — but `vcat(xi, xj .- xi)` is
    never materialised: the first Dense takes the two halves as the two segments of the MFMA kernel.
  * GatedGraphConv: W_l h on the MFMA kernel, propagate(copy_xj), the GRU cell = two contractions + one pointwise kernel.
  * DConv: every diffusion step `propagate(w_mul_xj, g | gᵀ, +; xj = T * deg)` is ONE launch (the degree scaling is the
    kernel's per-source factor); the two weight products of a step are the two segments of one dense call.
torch allocates; every arithmetic step is a libgnnmp call.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .graph import GNNGraph, check_num_edges, check_num_nodes, degree
from .layers import Dense, dense, glorot_uniform
from .msgpass import _gather, _scatter_plan, aggr_code

_CG_ACT = {None: L.ACT_IDENTITY, "identity": L.ACT_IDENTITY, "relu": L.ACT_RELU, "softplus": L.ACT_SOFTPLUS,
           "tanh": L.ACT_TANH}


def _add(a, b):
    out = torch.empty_like(a)
    L.check(L.load().gnnmp_add_f32(L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.stream_ptr()))
    return out


# ---------------------------------------------------------------------------------------------------------
# CGConv
# ---------------------------------------------------------------------------------------------------------
def cg_conv(l, g: GNNGraph, x, e=None):
    check_num_nodes(g, x)
    nin, ein = l.ch[0]
    out = l.ch[1]
    if e is not None:
        check_num_edges(g, e)
        assert e.shape[1] == ein
    else:
        assert ein == 0, "CGConv was built with edge features"
    x = x.contiguous()
    Wi, Wj, We, b = l.split_weights()
    fs_i = dense(x, Wi, b)                                  # [N][2 out]: (dense_f | dense_s) share of x_i, biases folded in
    fs_j = dense(x, Wj)
    fs_e = dense(e.contiguous(), We) if e is not None else None
    plan = g.plan(False)
    m = torch.empty((g.num_nodes, out), dtype=torch.float32, device=x.device)
    L.check(L.load().gnnmp_propagate_cg_f32(plan.handle, L.ptr(fs_i), L.ptr(fs_j), L.ptr(fs_e), _CG_ACT[l.act], L.ptr(m), out,
                                            L.stream_ptr()))
    if l.residual:
        if x.shape[1] == out:
            m = _add(m, x)
        else:
            import warnings
            warnings.warn("number of output features different from number of input features, residual not applied.")
    return m


class CGConv:
    """CGConv((in, ein) => out, act = identity; residual = false, bias = true): `ch` is `(in, out)` or `((in, ein), out)`"""

    takes_graph = True

    def __init__(self, ch, act=None, residual=False, bias=True, device="cuda", seed=None):
        cin, out = ch
        nin, ein = cin if isinstance(cin, (tuple, list)) else (cin, 0)
        assert act in _CG_ACT, f"unsupported activation {act!r}"
        self.ch, self.act, self.residual = ((nin, ein), out), act, bool(residual)
        sd = (lambda k: None if seed is None else seed + k)
        self.dense_f_weight = glorot_uniform(out, 2 * nin + ein, device=device, seed=sd(0))
        self.dense_s_weight = glorot_uniform(out, 2 * nin + ein, device=device, seed=sd(1))
        self.dense_f_bias = torch.zeros(out, dtype=torch.float32, device=device) if bias else None
        self.dense_s_bias = torch.zeros(out, dtype=torch.float32, device=device) if bias else None

    def split_weights(self):
        """([Wf_i; Ws_i], [Wf_j; Ws_j], [Wf_e; Ws_e] | None, [bf; bs] | None): the column blocks of dense_f / dense_s
        stacked so that one contraction gives both pre-activation shares (memory plumbing, cached per parameter version)"""
        ps = (self.dense_f_weight, self.dense_s_weight, self.dense_f_bias, self.dense_s_bias)
        key = tuple((p.data_ptr(), p._version) for p in ps if p is not None)
        if getattr(self, "_split_key", None) != key:
            (nin, ein), _ = self.ch
            Wf, Ws = self.dense_f_weight, self.dense_s_weight
            blk = lambda a, b: torch.cat([Wf[:, a:b], Ws[:, a:b]], dim=0).contiguous()
            bias = None if self.dense_f_bias is None else torch.cat([self.dense_f_bias, self.dense_s_bias]).contiguous()
            self._split = (blk(0, nin), blk(nin, 2 * nin), blk(2 * nin, 2 * nin + ein) if ein > 0 else None, bias)
            self._split_key = key
        return self._split

    def __call__(self, g, x, e=None):
        return cg_conv(self, g, x, e)


# ---------------------------------------------------------------------------------------------------------
# EdgeConv
# ---------------------------------------------------------------------------------------------------------
def edge_conv(l, g: GNNGraph, x):
    """propagate(edge_conv_message, g, aggr): l.nn(vcat(xi, xj .- xi)) per edge, then the aggregation"""
    check_num_nodes(g, x)
    x = x.contiguous()
    D = x.shape[1]
    xi = _gather(x, g.t, g.index_base)                      # (D, E) as the reference forms it
    dj = torch.empty((g.num_edges, D), dtype=torch.float32, device=x.device)
    L.check(L.load().gnnmp_edge_sub_f32(L.ptr(x), L.ptr(x), L.ptr(g.s), L.ptr(g.t), g.idx_bytes, g.index_base, g.num_edges,
                                        1, L.ptr(dj), D, L.stream_ptr()))                       # xj .- xi
    layers = l.nn if isinstance(l.nn, (list, tuple)) else [l.nn]
    first = layers[0]
    assert isinstance(first, Dense) and first.weight.shape[1] == 2 * D, "EdgeConv: nn must start with Dense(2 in => ...)"
    W = first.weight
    m = dense(xi, W[:, :D], first.bias, first.sigma, x2=dj, W2=W[:, D:])                       # nn's first layer on the vcat
    for layer in layers[1:]:
        m = layer(m)
    return _scatter_plan(l.aggr, m, g.plan(False))


class EdgeConv:
    """EdgeConv(nn; aggr = max) — `nn` a gnnmp Dense or a list of layers starting with Dense(2 in => ...)"""

    takes_graph = True

    def __init__(self, nn, aggr="max"):
        self.nn, self.aggr = nn, aggr

    def __call__(self, g, x):
        return edge_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# GatedGraphConv
# ---------------------------------------------------------------------------------------------------------
def gru_cell(m, h, Wi, Wh, b):
    """Flux.GRUCell: gates r, z, candidate from Wi m and Wh h (+ b), h' = (1 - z) h~ + z h"""
    gx, gh = dense(m, Wi), dense(h, Wh)
    out = torch.empty_like(h)
    L.check(L.load().gnnmp_gru_pointwise_f32(L.ptr(gx), L.ptr(gh), L.ptr(b), L.ptr(h), L.ptr(out), h.shape[0], h.shape[1],
                                             L.stream_ptr()))
    return out


def gated_graph_conv(l, g: GNNGraph, x):
    check_num_nodes(g, x)
    N, m = x.shape
    assert m <= l.dims, "number of input features must be less or equal to output features."
    if m < l.dims:
        h = torch.zeros((N, l.dims), dtype=torch.float32, device=x.device)      # vcat(x, zeros): a copy, no arithmetic
        h[:, :m] = x
    else:
        h = x.contiguous()
    plan = g.plan(False)
    lib = L.load()
    for i in range(l.num_layers):
        mm = dense(h, l.weight[i])
        agg = torch.empty_like(mm)
        L.check(lib.gnnmp_propagate_f32(plan.handle, L.COPY_XJ, aggr_code(l.aggr), L.ptr(mm), None, None, None, L.ptr(agg),
                                        l.dims, L.stream_ptr()))
        h = gru_cell(agg, h, l.gru_Wi, l.gru_Wh, l.gru_b)
    return h


class GatedGraphConv:
    """GatedGraphConv(out, num_layers; aggr = +): weight [num_layers][out][out]; the GRU cell's Wi, Wh [3 out][out], b [3 out]"""

    takes_graph = True

    def __init__(self, dims, num_layers, aggr="+", device="cuda", seed=None):
        sd = (lambda k: None if seed is None else seed + k)
        self.dims, self.num_layers, self.aggr = int(dims), int(num_layers), aggr
        self.weight = torch.stack([glorot_uniform(dims, dims, device=device, seed=sd(i)) for i in range(num_layers)]).contiguous()
        self.gru_Wi = glorot_uniform(3 * dims, dims, device=device, seed=sd(100))
        self.gru_Wh = glorot_uniform(3 * dims, dims, device=device, seed=sd(101))
        self.gru_b = torch.zeros(3 * dims, dtype=torch.float32, device=device)

    def __call__(self, g, x):
        return gated_graph_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# DConv
# ---------------------------------------------------------------------------------------------------------
def d_conv(l, g: GNNGraph, x):
    check_num_nodes(g, x)
    from .backward import plan_transposed
    x = x.contiguous()
    lib = L.load()
    deg_out, deg_in = degree(g, torch.float32, dir="out"), degree(g, torch.float32, dir="in")
    fplan, bplan = g.plan(False), plan_transposed(g, False)
    msg = L.COPY_XJ if g.w is None else L.W_MUL_XJ
    D = x.shape[1]

    def step(plan, T, deg):          # propagate(w_mul_xj, g | gt, +; xj = T * Diagonal(deg)): the scaling is the source factor
        out = torch.empty_like(T)
        L.check(lib.gnnmp_propagate_f32(plan.handle, msg, L.SUM, L.ptr(T), L.ptr(g.w), L.ptr(deg), None, L.ptr(out), D,
                                        L.stream_ptr()))
        return out

    def cheb(P, T0):                 # 2 * P - T0: the doubling is exact, so (P + P) - T0 rounds once like the reference
        out = torch.empty_like(P)
        L.check(lib.gnnmp_axpy_f32(-1.0, L.ptr(T0), L.ptr(_add(P, P)), L.ptr(out), P.numel(), L.stream_ptr()))
        return out

    W = l.weights
    h = dense(x, W[0, 0], None, None, x2=x, W2=W[1, 0])
    T0 = x
    T1_in = T1_out = None
    if l.k > 1:
        T1_out, T1_in = step(fplan, T0, deg_out), step(bplan, T0, deg_in)
        h = _add(h, dense(T1_in, W[0, 1], None, None, x2=T1_out, W2=W[1, 1]))
    for i in range(1, l.k):          # the reference's `for i in 2:l.k` reads weight slice i (1-based): slice 2 twice
        T2_in, T2_out = cheb(step(bplan, T1_in, deg_in), T0), cheb(step(fplan, T1_out, deg_out), T0)
        h = _add(h, dense(T2_in, W[0, i], None, None, x2=T2_out, W2=W[1, i]))
        T1_in, T1_out = T2_in, T2_out
    if l.bias is not None:
        from .layers import bias_act
        h = bias_act(h, l.bias, None)
    return h


class DConv:
    """DConv(in => out, k; bias = true): weights [2][k][out][in] (Julia (2, k, out, in))"""

    takes_graph = True

    def __init__(self, ch, k, bias=True, device="cuda", seed=None):
        cin, out = ch
        self.cin, self.out, self.k = cin, out, int(k)
        sd = (lambda j: None if seed is None else seed + j)
        self.weights = torch.stack([torch.stack([glorot_uniform(out, cin, device=device, seed=sd(a * 64 + i))
                                                 for i in range(k)]) for a in range(2)]).contiguous()
        self.bias = torch.zeros(out, dtype=torch.float32, device=device) if bias else None

    def __call__(self, g, x):
        return d_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# NNConv
# ---------------------------------------------------------------------------------------------------------
def nn_conv(l, g: GNNGraph, x, e):
    """conv.jl:260-273: m = propagate(W_k x_j with W_k = reshape(nn(e)_k, out, in), aggr); σ.(weight * x .+ m .+ bias).
    nn(e) is formed once ([E][out * in], as the reference does); the per-edge matrix-vector products and the aggregation
    are one pass (gnnmp_propagate_nn_f32) — the (out, E) message array is never built."""
    check_num_nodes(g, x)
    check_num_edges(g, e)
    x = x.contiguous()
    out, nin = l.weight.shape
    assert x.shape[1] == nin
    we = e.contiguous()
    for layer in (l.nn if isinstance(l.nn, (list, tuple)) else [l.nn]):
        we = layer(we)
    assert we.shape == (g.num_edges, out * nin), "NNConv: nn must map edge features to out * in channels"
    m = torch.empty((g.num_nodes, out), dtype=torch.float32, device=x.device)
    L.check(L.load().gnnmp_propagate_nn_f32(g.plan(False).handle, aggr_code(l.aggr), L.ptr(x), L.ptr(we.contiguous()), L.ptr(m),
                                            nin, out, L.stream_ptr()))
    from .layers import bias_act
    return bias_act(_add(dense(x, l.weight), m), l.bias, l.sigma)


class NNConv:
    """NNConv(in => out, nn, σ = identity; aggr = +, bias = true) — GraphNeuralNetworks/src/layers/conv.jl (edge-conditioned
    convolution); `nn`: a gnnmp Dense or list of layers mapping edge features to out * in channels"""

    takes_graph = True

    def __init__(self, ch, nn, sigma=None, aggr="+", bias=True, device="cuda", seed=None):
        cin, out = ch
        self.nn, self.sigma, self.aggr = nn, sigma, aggr
        self.weight = glorot_uniform(out, cin, device=device, seed=seed)
        self.bias = torch.zeros(out, dtype=torch.float32, device=device) if bias else None

    def __call__(self, g, x, e):
        return nn_conv(self, g, x, e)


# ---------------------------------------------------------------------------------------------------------
# MEGNetConv
# ---------------------------------------------------------------------------------------------------------
def megnet_conv(l, g: GNNGraph, x, e):
    """conv.jl:356-368: ē = ϕe(vcat(xi, xj, e)), xᵉ = aggregate_neighbors(g, aggr, ē), x̄ = ϕv(vcat(x, xᵉ)) -> (x̄, ē).
    Neither vcat is materialised: the first Dense of each chain takes its blocks as separate contractions."""
    from .layers import bias_act
    check_num_nodes(g, x)
    check_num_edges(g, e)
    x, e = x.contiguous(), e.contiguous()
    n_in, n_e = x.shape[1], e.shape[1]
    pe, pv = list(l.phi_e), list(l.phi_v)
    first = pe[0]
    assert isinstance(first, Dense) and first.weight.shape[1] == 2 * n_in + n_e, "MEGNetConv: ϕe must start with Dense(2 in + ein => ...)"
    xi, xj = _gather(x, g.t, g.index_base), _gather(x, g.s, g.index_base)
    W = first.weight
    eb = _add(dense(xi, W[:, :n_in], None, None, x2=xj, W2=W[:, n_in:2 * n_in]), dense(e, W[:, 2 * n_in:]))
    eb = bias_act(eb, first.bias, first.sigma)
    for layer in pe[1:]:
        eb = layer(eb)
    xe = _scatter_plan(l.aggr, eb, g.plan(False))
    fv = pv[0]
    assert isinstance(fv, Dense) and fv.weight.shape[1] == n_in + xe.shape[1], "MEGNetConv: ϕv must start with Dense(in + out => ...)"
    xb = dense(x, fv.weight[:, :n_in], fv.bias, fv.sigma, x2=xe, W2=fv.weight[:, n_in:])
    for layer in pv[1:]:
        xb = layer(xb)
    return xb, eb


class MEGNetConv:
    """MEGNetConv(in => out; aggr = mean) or MEGNetConv(ϕe, ϕv; aggr = mean) with ϕe / ϕv lists of gnnmp Dense layers
    (default: Dense(3 in => out, relu), Dense(out => out) and Dense(in + out => out, relu), Dense(out => out))"""

    takes_graph = True

    def __init__(self, ch=None, phi_e=None, phi_v=None, aggr="mean", device="cuda", seed=None):
        if ch is not None:
            nin, nout = ch
            sd = (lambda k: None if seed is None else seed + k)
            phi_e = [Dense((3 * nin, nout), "relu", device=device, seed=sd(0)), Dense((nout, nout), None, device=device, seed=sd(1))]
            phi_v = [Dense((nin + nout, nout), "relu", device=device, seed=sd(2)), Dense((nout, nout), None, device=device, seed=sd(3))]
        self.phi_e, self.phi_v, self.aggr = phi_e, phi_v, aggr

    def __call__(self, g, x, e):
        return megnet_conv(self, g, x, e)


# ---------------------------------------------------------------------------------------------------------
# GMMConv
# ---------------------------------------------------------------------------------------------------------
def gmm_conv(l, g: GNNGraph, x, e):
    """conv.jl:372-401: Gaussian mixture weights per edge and kernel, `propagate(e_mul_xj, g, mean)` of the K blocks of
    dense_x(x), mean over the kernels, σ.(m .+ bias), optional residual"""
    check_num_nodes(g, x)
    (nin, ein), out = l.ch
    assert e.shape[1] == ein and e.shape[0] == g.num_edges, "Pseudo-cordinate dimension is not equal to (ein,num_edge)"
    x, e = x.contiguous(), e.contiguous()
    lib = L.load()
    K = l.K
    w = torch.empty((g.num_edges, K * out), dtype=torch.float32, device=x.device)
    L.check(lib.gnnmp_gmm_weights_f32(L.ptr(e), L.ptr(l.mu), L.ptr(l.sigma_inv), L.ptr(w), g.num_edges, ein, K, out, L.stream_ptr()))
    xj = dense(x, l.dense_x_weight)                           # [N][K * out]: kernel k's block at k * out
    m = torch.empty((g.num_nodes, K * out), dtype=torch.float32, device=x.device)
    L.check(lib.gnnmp_propagate_emul_f32(g.plan(False).handle, L.MEAN, L.ptr(xj), L.ptr(w), L.ptr(m), K * out, L.stream_ptr()))
    y = torch.empty((g.num_nodes, out), dtype=torch.float32, device=x.device)
    from .layers import _act_code
    code, post = _act_code(l.sigma)
    L.check(lib.gnnmp_head_mean_f32(L.ptr(m), L.ptr(l.bias), code, L.ptr(y), g.num_nodes, K, out, L.stream_ptr()))
    if post is not None:
        y = post(y)
    if l.residual:
        if x.shape[1] == out:
            y = _add(y, x)
        else:
            import warnings
            warnings.warn("Residual not applied : output feature is not equal to input_feature")
    return y


class GMMConv:
    """GMMConv((in, ein) => out, σ = identity; K = 1, bias = true, residual = false): mu, sigma_inv [K][ein] (Julia (ein, K))"""

    takes_graph = True

    def __init__(self, ch, sigma=None, K=1, bias=True, residual=False, device="cuda", seed=None):
        (nin, ein), out = ch
        sd = (lambda k: None if seed is None else seed + k)
        self.ch, self.sigma, self.K, self.residual = ((nin, ein), out), sigma, int(K), bool(residual)
        self.mu = glorot_uniform(K, ein, device=device, seed=sd(0))
        self.sigma_inv = glorot_uniform(K, ein, device=device, seed=sd(1))
        self.bias = torch.zeros(out, dtype=torch.float32, device=device) if bias else None
        self.dense_x_weight = glorot_uniform(out * K, nin, device=device, seed=sd(2))

    def __call__(self, g, x, e):
        return gmm_conv(self, g, x, e)


# ---------------------------------------------------------------------------------------------------------
# EGNNConv
# ---------------------------------------------------------------------------------------------------------
_ACT_CODES = {None: L.ACT_IDENTITY, "identity": L.ACT_IDENTITY, "relu": L.ACT_RELU, "softplus": L.ACT_SOFTPLUS,
              "tanh": L.ACT_TANH, "swish": L.ACT_SWISH}


def _bias_act_code(x, bias, act):
    out = torch.empty_like(x)
    b = None if bias is None else bias.contiguous()
    L.check(L.load().gnnmp_bias_act_f32(L.ptr(x), L.ptr(b), _ACT_CODES[act], L.ptr(out), x.shape[0], x.shape[1], L.stream_ptr()))
    return out


def _chain(layers, z):
    """a chain of (weight, bias, act) Dense layers, activations from the gnnmp_act set (swish included)"""
    for W, b, act in layers:
        z = _bias_act_code(dense(z, W), b, act)
    return z


def egnn_conv(l, g: GNNGraph, h, x, e=None):
    """conv.jl:459-495 -> (h', x'): radial feature and normalised coordinate difference per edge, ϕe on
    vcat(h_i, h_j, |x_i - x_j|², e) without the vcat, ϕx scaling the differences, sum / mean aggregation, ϕh on (h, Σ)."""
    check_num_nodes(g, h)
    nin, ein = l.num_features["in"], l.num_features["edge"]
    if ein > 0:
        assert e is not None, "Edge features must be provided."
    assert h.shape[1] == nin, "Input features must match layer input size."
    h, x = h.contiguous(), x.contiguous()
    lib = L.load()
    E, Dx = g.num_edges, x.shape[1]
    xd = torch.empty((E, Dx), dtype=torch.float32, device=x.device)
    L.check(lib.gnnmp_edge_sub_f32(L.ptr(x), L.ptr(x), L.ptr(g.s), L.ptr(g.t), g.idx_bytes, g.index_base, E, 0, L.ptr(xd), Dx,
                                   L.stream_ptr()))                                       # xi_sub_xj
    sq = torch.empty((E, 1), dtype=torch.float32, device=x.device)
    xn = torch.empty_like(xd)
    L.check(lib.gnnmp_row_sqnorm_normalize_f32(L.ptr(xd), L.ptr(sq), L.ptr(xn), 1e-6, E, Dx, L.stream_ptr()))
    hi, hj = _gather(h, g.t, g.index_base), _gather(h, g.s, g.index_base)
    (W0, b0, a0), rest = l.phi_e[0], l.phi_e[1:]
    assert W0.shape[1] == 2 * nin + 1 + ein
    tail = sq if ein == 0 else torch.cat([sq, e.contiguous()], dim=1)                     # (1 + ein, E): memory plumbing
    pre = _add(dense(hi, W0[:, :nin], None, None, x2=hj, W2=W0[:, nin:2 * nin]), dense(tail, W0[:, 2 * nin:]))
    mh = _chain(rest, _bias_act_code(pre, b0, a0))
    mx = _chain(l.phi_x, mh)                                                              # (1, E)
    mxd = torch.empty_like(xn)
    L.check(lib.gnnmp_mul_rows_f32(L.ptr(mx), 1, L.ptr(xn), L.ptr(mxd), E, Dx, L.stream_ptr()))
    plan = g.plan(False)
    h_aggr, x_aggr = _scatter_plan("+", mh, plan), _scatter_plan("mean", mxd, plan)
    (Wh, bh, ah), resth = l.phi_h[0], l.phi_h[1:]
    hn = _chain(resth, _bias_act_code(dense(h, Wh[:, :nin], None, None, x2=h_aggr, W2=Wh[:, nin:]), bh, ah))
    return (_add(h, hn) if l.residual else hn), _add(x, x_aggr)


class EGNNConv:
    """EGNNConv((in, ein) => out; hidden_size = 2 in, residual = false): ϕe, ϕx, ϕh as lists of (weight, bias, act) with the
    reference's shapes and swish activations (GraphNeuralNetworks/src/layers/conv.jl:1364-1385)"""

    takes_graph = True

    def __init__(self, ch, hidden_size=None, residual=False, device="cuda", seed=None):
        cin, out = ch
        nin, ein = cin if isinstance(cin, (tuple, list)) else (cin, 0)
        hid = 2 * nin if hidden_size is None else int(hidden_size)
        if residual:
            assert nin == out, "Residual connection only possible if in_size == out_size"
        sd = iter(range(0 if seed is None else seed, 10 ** 9))
        z = lambda k: torch.zeros(k, dtype=torch.float32, device=device)
        mk = lambda o, i, act, bias=True: (glorot_uniform(o, i, device=device, seed=next(sd)), z(o) if bias else None, act)
        self.phi_e = [mk(hid, 2 * nin + ein + 1, "swish"), mk(hid, hid, "swish")]
        self.phi_h = [mk(hid, nin + hid, "swish"), mk(out, hid, None)]
        self.phi_x = [mk(hid, hid, "swish"), mk(1, hid, None, bias=False)]
        self.num_features = {"in": nin, "edge": ein, "out": out, "hidden": hid}
        self.residual = bool(residual)

    def __call__(self, g, h, x, e=None):
        return egnn_conv(self, g, h, x, e)


# ---------------------------------------------------------------------------------------------------------
# ChebConv
# ---------------------------------------------------------------------------------------------------------
def _colsum1(v):
    """deterministic sum of an (N, 1) column on the device -> python float"""
    from .backward import dense_grad_w
    return float(dense_grad_w(v, v, need_w=False)[1][0])


def _vdot(a, b):
    p = torch.empty_like(a)
    L.check(L.load().gnnmp_mul_rows_f32(L.ptr(a), 1, L.ptr(b), L.ptr(p), a.shape[0], 1, L.stream_ptr()))
    return _colsum1(p)


def _axpy(alpha, x, y):
    out = torch.empty_like(x)
    L.check(L.load().gnnmp_axpy_f32(float(alpha), L.ptr(x), L.ptr(y), L.ptr(out), x.numel(), L.stream_ptr()))
    return out


def scaled_laplacian_op(g: GNNGraph, steps: int = 64, seed: int = 0):
    """(c, ss_slot, w_slot, λmax) of scaled_laplacian(g) (GNNGraphs/src/query.jl:442-479) for an undirected graph, never as a
    matrix: Ã v is one fused hop (out-degree normalisation as the kernel's source / destination factors), λmax = eigmax(I - Ã)
    by `steps` Lanczos iterations whose vector work (hop, dot, axpy) runs on the device; the recurrence scalars and the
    eigenvalues of the small tridiagonal matrix are host float64 (the reference's own eigsolve, KrylovKit, is host code too).
    Cached on the graph."""
    import numpy as np
    from .layers import _inv_sqrt
    from .sampling import is_bidirected
    key = ("cheb", None if g.w is None else (g.w.data_ptr(), g.w._version))
    hit = g._cache.get(key)
    if hit is not None:
        return hit
    assert is_bidirected(g), "ChebConv: the scaled Laplacian is taken of an undirected (bidirected) graph"
    lib = L.load()
    plan = g.plan(False)
    d = degree(g, torch.float32, dir="out")
    assert bool((d != 0).all()), "Graph contains isolated nodes, cannot compute `normalized_adjacency`."
    c = _inv_sqrt(d)
    ss = torch.empty(plan.n_total, dtype=torch.float32, device=g.device)
    L.check(lib.gnnmp_plan_slot_gather_f32(plan.handle, 0, L.ptr(c), L.ptr(ss), L.stream_ptr()))
    ws = None
    if g.w is not None:
        ws = torch.empty(plan.n_total, dtype=torch.float32, device=g.device)
        L.check(lib.gnnmp_plan_slot_gather_f32(plan.handle, 1, L.ptr(g.w), L.ptr(ws), L.stream_ptr()))
    N = g.num_nodes

    def lap(v):                                             # L v = v - Ã v
        av = torch.empty_like(v)
        L.check(lib.gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(v), L.ptr(ws), L.ptr(ss), L.ptr(c), L.ptr(av), 1,
                                              L.stream_ptr()))
        return _axpy(-1.0, av, v)

    gen = torch.Generator(device="cpu").manual_seed(seed)
    zero = torch.zeros((N, 1), dtype=torch.float32, device=g.device)
    v = torch.randn((N, 1), generator=gen, dtype=torch.float32).to(g.device)
    v = _axpy(1.0 / np.sqrt(_vdot(v, v)), v, zero)
    vp, beta = zero, 0.0
    al, be = [], []
    for j in range(min(steps, N)):
        w = lap(v)
        a = _vdot(w, v)
        w = _axpy(-a, v, w)
        if j > 0:
            w = _axpy(-beta, vp, w)
        al.append(a)
        beta = float(np.sqrt(max(_vdot(w, w), 0.0)))
        if beta < 1e-6:
            break
        be.append(beta)
        vp, v = v, _axpy(1.0 / beta, w, zero)
    k = len(al)
    T = np.diag(np.array(al, np.float64)) + np.diag(np.array(be[:k - 1], np.float64), 1) + np.diag(np.array(be[:k - 1], np.float64), -1)
    lam = float(np.linalg.eigvalsh(T)[-1])
    g._cache[key] = (c, ss, ws, lam)
    return g._cache[key]


def cheb_conv(l, g: GNNGraph, x):
    """conv.jl:83-98: Z_1 = X, Z_2 = X L̃, Z_k = 2 Z_{k-1} L̃ - Z_{k-2}, Y = Σ W_k Z_k .+ bias with L̃ = 2/λmax (I - Ã) - I
    applied as (2/λmax - 1) Z - (2/λmax) Ã Z: one fused hop + two axpy per order"""
    check_num_nodes(g, x)
    assert x.shape[1] == l.weight.shape[2], "Input feature size must match input channel size."
    c, ss, ws, lam = scaled_laplacian_op(g)
    lib = L.load()
    plan = g.plan(False)
    x = x.contiguous()

    def times_L(Z):
        az = torch.empty_like(Z)
        L.check(lib.gnnmp_propagate_slots_f32(plan.handle, L.SUM, L.ptr(Z), L.ptr(ws), L.ptr(ss), L.ptr(c), L.ptr(az), Z.shape[1],
                                              L.stream_ptr()))
        zero = torch.zeros_like(Z)
        return _axpy(-2.0 / lam, az, _axpy(2.0 / lam - 1.0, Z, zero))

    Zp, Z = x, None
    Y = dense(Zp, l.weight[0])
    if l.k > 1:
        Z = times_L(x)
        Y = _add(Y, dense(Z, l.weight[1]))
    for i in range(2, l.k):
        LZ = times_L(Z)
        Z, Zp = _axpy(-1.0, Zp, _add(LZ, LZ)), Z
        Y = _add(Y, dense(Z, l.weight[i]))
    if l.bias is not None:
        from .layers import bias_act
        Y = bias_act(Y, l.bias, None)
    return Y


class ChebConv:
    """ChebConv(in => out, k; bias = true): weight [k][out][in] (Julia (out, in, k))"""

    takes_graph = True

    def __init__(self, ch, k, bias=True, device="cuda", seed=None):
        cin, out = ch
        self.k = int(k)
        sd = (lambda j: None if seed is None else seed + j)
        self.weight = torch.stack([glorot_uniform(out, cin, device=device, seed=sd(i)) for i in range(k)]).contiguous()
        self.bias = torch.zeros(out, dtype=torch.float32, device=device) if bias else None

    def __call__(self, g, x):
        return cheb_conv(self, g, x)


# ---------------------------------------------------------------------------------------------------------
# Set2Set
# ---------------------------------------------------------------------------------------------------------
def set2set_pool(l, g: GNNGraph, x):
    """GNNlib/src/layers/pool.jl:31-44: num_iters rounds of LSTM query -> broadcast_nodes -> softmax_nodes of q . x ->
    reduce_nodes(+, x .* α) -> q* = vcat(q, r); returns [num_graphs][2 n_in]"""
    from .utils import broadcast_nodes, reduce_nodes, softmax_nodes
    check_num_nodes(g, x)
    x = x.contiguous()
    N, n_in = x.shape
    G = g.num_graphs
    lib = L.load()
    z = lambda r, c: torch.zeros((r, c), dtype=torch.float32, device=x.device)
    q, r, h, c = z(G, n_in), z(G, n_in), z(G, n_in), z(G, n_in)
    for _ in range(l.num_iters):
        gx = dense(q, l.Wi[:, :n_in], None, None, x2=r, W2=l.Wi[:, n_in:])       # Wi * vcat(q, r) without the vcat
        gh = dense(h, l.Wh)
        hn, cn = torch.empty_like(h), torch.empty_like(c)
        L.check(lib.gnnmp_lstm_pointwise_f32(L.ptr(gx), L.ptr(gh), L.ptr(l.b), L.ptr(c), L.ptr(hn), L.ptr(cn), G, n_in,
                                             L.stream_ptr()))
        h, c = hn, cn
        q = h
        qn = broadcast_nodes(g, q)
        sc = torch.empty((N, 1), dtype=torch.float32, device=x.device)
        L.check(lib.gnnmp_rowdot_f32(L.ptr(qn), L.ptr(x), L.ptr(sc), N, n_in, L.stream_ptr()))
        alpha = softmax_nodes(g, sc)
        xa = torch.empty_like(x)
        L.check(lib.gnnmp_mul_rows_f32(L.ptr(alpha), 1, L.ptr(x), L.ptr(xa), N, n_in, L.stream_ptr()))
        r = reduce_nodes("+", g, xa)
    return torch.cat([q, r], dim=1)                                               # vcat(q, r): memory plumbing


class Set2Set:
    """Set2Set(n_in, n_iters, n_layers = 1): LSTMCell(2 n_in => n_in) — Wi [4 n_in][2 n_in], Wh [4 n_in][n_in], b [4 n_in]"""

    takes_graph = True

    def __init__(self, n_in, n_iters, n_layers=1, device="cuda", seed=None):
        assert n_layers == 1, "multiple layers not implemented yet"
        sd = (lambda k: None if seed is None else seed + k)
        self.num_iters = int(n_iters)
        self.Wi = glorot_uniform(4 * n_in, 2 * n_in, device=device, seed=sd(0))
        self.Wh = glorot_uniform(4 * n_in, n_in, device=device, seed=sd(1))
        self.b = torch.zeros(4 * n_in, dtype=torch.float32, device=device)

    def __call__(self, g, x):
        return set2set_pool(self, g, x)
