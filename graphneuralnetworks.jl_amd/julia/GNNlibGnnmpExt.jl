# GNNlibGnnmpExt.jl — the reference-side binding of libgnnmp.so (include/gnnmp.h).
#
# This is the file a GNNlib.jl maintainer adds next to GNNlib/ext/GNNlibAMDGPUExt.jl (declare `Gnnmp_jll` or a path to
# libgnnmp.so plus `AMDGPU` as weakdeps in GNNlib/Project.toml:17-23, extension = ["AMDGPU"]).  It replaces the three
# methods of GNNlibAMDGPUExt.jl:13-32 — which today only *disable* the SpMM fast path on ROCm arrays and fall back to
# gather -> message -> atomic scatter — with calls into the fused HIP kernels, and adds the fused GCN / GAT / pooling
# fast paths.  Julia is NOT available in the build container, so this file is written against the reference's sources
# but has never been executed; it is kept deliberately thin (every method is a size check + one @ccall) and it mirrors
# 1:1 the ctypes host layer in graphneuralnetworks.jl_amd/gnnmp/, which IS tested on an MI355X.
#
# Layout contract: Julia's column-major (D, N) feature matrix is exactly the C row-major [N][D] the library wants, and
# GNNGraph's COO vectors (1-based Int64, or Int32) are passed untouched (idx_bytes, index_base = 1).

module GNNlibGnnmpExt

using AMDGPU: AMDGPU, ROCArray, ROCVector, ROCMatrix, AnyROCMatrix
using GNNlib: GNNlib, propagate, copy_xj, e_mul_xj, w_mul_xj
using GNNGraphs: GNNGraphs, GNNGraph, COO_T, edge_index, get_edge_weight, check_num_nodes, check_num_edges
using Statistics: mean

const libgnnmp = get(ENV, "GNNMP_LIB", "libgnnmp.so")

# ---- status / stream plumbing ---------------------------------------------------------------------------------------
struct GnnmpError <: Exception
    status::Cint
    msg::String
end
function check(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(@ccall libgnnmp.gnnmp_last_error()::Cstring)
    throw(GnnmpError(status, msg))
end
stream_ptr() = Base.unsafe_convert(Ptr{Cvoid}, AMDGPU.stream())   # launch on AMDGPU.jl's task-local stream, no syncs
devptr(x::ROCArray) = Base.unsafe_convert(Ptr{Cvoid}, pointer(x))
devptr(::Nothing) = C_NULL

const SUM, MEAN, MAX, MIN = Cint(0), Cint(1), Cint(2), Cint(3)
aggr_code(::typeof(+)) = SUM
aggr_code(::typeof(mean)) = MEAN
aggr_code(::typeof(max)) = MAX
aggr_code(::typeof(min)) = MIN
const FusedAggr = Union{typeof(+), typeof(mean), typeof(max), typeof(min)}

# ---- plan cache: one dst-sorted CSR per (s, t) pair, keyed by the identity of the index vectors -----------------------
mutable struct Plan
    handle::Ptr{Cvoid}
    function Plan(s::ROCVector{I}, t::ROCVector{I}, n_src::Int, n_dst::Int, self_loops::Bool) where {I <: Union{Int32, Int64}}
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(@ccall libgnnmp.gnnmp_plan_create(h::Ptr{Ptr{Cvoid}}, devptr(s)::Ptr{Cvoid}, devptr(t)::Ptr{Cvoid},
                                                sizeof(I)::Cint, 1::Cint, n_src::Int64, n_dst::Int64,
                                                length(s)::Int64, self_loops::Cint, 0::Cint,
                                                stream_ptr()::Ptr{Cvoid})::Cint)
        p = new(h[])
        finalizer(p -> (@ccall libgnnmp.gnnmp_plan_destroy(p.handle::Ptr{Cvoid})::Cint), p)
        return p
    end
end
const PLANS = WeakKeyDict{Any, Dict{Bool, Plan}}()   # GNNGraph is immutable and shares s, t across copies (gnngraph.jl:187-211)
function plan(g::GNNGraph{<:COO_T}; self_loops::Bool = false)
    s, t = edge_index(g)
    d = get!(() -> Dict{Bool, Plan}(), PLANS, s)
    get!(() -> Plan(s, t, g.num_nodes, g.num_nodes, self_loops), d, self_loops)
end

# ---- propagate: replaces GNNlib/ext/GNNlibAMDGPUExt.jl:13-32 (and extends it to mean / max / min) ----------------------
function fused_propagate(g::GNNGraph, aggr, xj::AnyROCMatrix{Float32}, w; self_loops = false, scale_src = nothing,
                         scale_dst = nothing)
    check_num_nodes(g, xj)                                        # AssertionError contract (GNNGraphs/src/utils.jl:1-28)
    p = plan(g; self_loops)
    D = size(xj, 1)
    out = similar(xj, D, g.num_nodes)
    msg = w === nothing ? Cint(0) : Cint(1)
    check(@ccall libgnnmp.gnnmp_propagate_f32(p.handle::Ptr{Cvoid}, msg::Cint, aggr_code(aggr)::Cint,
                                              devptr(xj)::Ptr{Cvoid}, devptr(w)::Ptr{Cvoid},
                                              devptr(scale_src)::Ptr{Cvoid}, devptr(scale_dst)::Ptr{Cvoid},
                                              devptr(out)::Ptr{Cvoid}, D::Int64, stream_ptr()::Ptr{Cvoid})::Cint)
    return out
end

function GNNlib.propagate(::typeof(copy_xj), g::GNNGraph{<:COO_T}, aggr::FusedAggr, xi, xj::AnyROCMatrix{Float32}, e)
    fused_propagate(g, aggr, xj, nothing)
end
function GNNlib.propagate(::typeof(e_mul_xj), g::GNNGraph{<:COO_T}, aggr::FusedAggr, xi, xj::AnyROCMatrix{Float32},
                          e::ROCVector{Float32})
    check_num_edges(g, e)
    fused_propagate(g, aggr, xj, e)
end
function GNNlib.propagate(::typeof(w_mul_xj), g::GNNGraph{<:COO_T}, aggr::FusedAggr, xi, xj::AnyROCMatrix{Float32},
                          e::Nothing)
    fused_propagate(g, aggr, xj, get_edge_weight(g))              # the CPU fast path's weighted adjacency (msgpass.jl:234-238)
end

# ---- leaf ops for arbitrary closures: GNNGraphs/src/gatherscatter.jl:4,12-18 ------------------------------------------
function GNNGraphs._gather(x::ROCArray{Float32}, i::ROCVector{I}) where {I <: Union{Int32, Int64}}
    D = prod(size(x)[1:(end - 1)])
    out = similar(x, size(x)[1:(end - 1)]..., length(i))
    check(@ccall libgnnmp.gnnmp_gather_f32(devptr(x)::Ptr{Cvoid}, devptr(i)::Ptr{Cvoid}, sizeof(I)::Cint, 1::Cint,
                                           length(i)::Int64, devptr(out)::Ptr{Cvoid}, D::Int64,
                                           stream_ptr()::Ptr{Cvoid})::Cint)
    return out
end
# deterministic, edge-order scatter through the plan of the graph's targets; used by aggregate_neighbors
function GNNlib.aggregate_neighbors(g::GNNGraph{<:COO_T}, aggr::FusedAggr, m::ROCArray{Float32})
    check_num_edges(g, m)
    p = plan(g)
    D = prod(size(m)[1:(end - 1)])
    out = similar(m, size(m)[1:(end - 1)]..., g.num_nodes)
    check(@ccall libgnnmp.gnnmp_scatter_f32(p.handle::Ptr{Cvoid}, aggr_code(aggr)::Cint, devptr(m)::Ptr{Cvoid},
                                            devptr(out)::Ptr{Cvoid}, D::Int64, stream_ptr()::Ptr{Cvoid})::Cint)
    return out
end

# ---- softmax_edge_neighbors: GNNlib/src/utils.jl:84-97 ----------------------------------------------------------------
function GNNlib.softmax_edge_neighbors(g::GNNGraph{<:COO_T}, e::ROCArray{Float32})
    @assert size(e)[end] == g.num_edges
    H = prod(size(e)[1:(end - 1)])
    out = similar(e)
    check(@ccall libgnnmp.gnnmp_edge_softmax_f32(plan(g).handle::Ptr{Cvoid}, devptr(e)::Ptr{Cvoid},
                                                 devptr(out)::Ptr{Cvoid}, H::Int64, stream_ptr()::Ptr{Cvoid})::Cint)
    return out
end

# ---- reduce_nodes / global_pool: GNNlib/src/utils.jl:12-16 (batch-built indicators are sorted) ---------------------------
function GNNlib.reduce_nodes(aggr::FusedAggr, g::GNNGraph, x::AnyROCMatrix{Float32})
    @assert size(x)[end] == g.num_nodes
    gi = GNNGraphs.graph_indicator(g)
    D = size(x, 1)
    out = similar(x, D, g.num_graphs)
    check(@ccall libgnnmp.gnnmp_segment_pool_f32(aggr_code(aggr)::Cint, devptr(x)::Ptr{Cvoid}, devptr(gi)::Ptr{Cvoid},
                                                 sizeof(eltype(gi))::Cint, 1::Cint, devptr(out)::Ptr{Cvoid}, D::Int64,
                                                 g.num_nodes::Int64, g.num_graphs::Int64,
                                                 stream_ptr()::Ptr{Cvoid})::Cint)
    return out
end

# ---- fused GATConv attention path: GNNlib/src/layers/conv.jl:112-150 with e === nothing, dropout = 0 --------------------
# (called from a gat_conv method specialised on ROCArray inputs: Wx = reshape(l.dense_x(x), C, H, N) is the (C*H, N)
#  matrix as stored, `a` is l.a as stored — see the NOTE below.)
function gat_attention(g::GNNGraph{<:COO_T}, Wx::AnyROCMatrix{Float32}, a::ROCMatrix{Float32}, slope::Float32,
                       bias, relu::Bool, heads::Int, chout::Int; self_loops::Bool)
    out = similar(Wx)
    check(@ccall libgnnmp.gnnmp_gat_conv_f32(plan(g; self_loops).handle::Ptr{Cvoid}, devptr(Wx)::Ptr{Cvoid},
                                             C_NULL::Ptr{Cvoid}, devptr(a)::Ptr{Cvoid}, slope::Cfloat,
                                             devptr(bias)::Ptr{Cvoid}, Cint(relu)::Cint, devptr(out)::Ptr{Cvoid},
                                             heads::Int64, chout::Int64, stream_ptr()::Ptr{Cvoid})::Cint)
    return out
end
# NOTE on `a`: Julia stores l.a of size (2C, H) column-major = C row-major [H][2C]: exactly the layout gnnmp.h asks for,
# so `l.a` itself is passed — no permutedims.

# ---- the other attention layers on the same one-pass kernel (gnnmp_attn_conv_f32) -----------------------------------------
# mode 1: gatv2_conv (conv.jl:171-214; Q = Wxi, K = Wxj, a = l.a (C, H) as stored)   mode 2: transformer_conv attention
# (conv.jl:553-616; Q = W3x, K = W4x, V = W2x, scale = l.sqrt_out)   mode 3: agnn_conv (conv.jl:337-352; K = x, scale = l.β[1])
function attention(g::GNNGraph{<:COO_T}, mode::Int, K::AnyROCMatrix{Float32}; Q = nothing, V = nothing, a = nothing,
                   slope = 0.2f0, scale = 1f0, bias = nothing, relu = false, heads::Int = 1, self_loops::Bool)
    out = similar(K)
    check(@ccall libgnnmp.gnnmp_attn_conv_f32(plan(g; self_loops).handle::Ptr{Cvoid}, mode::Cint, devptr(Q)::Ptr{Cvoid},
                                              devptr(K)::Ptr{Cvoid}, devptr(V)::Ptr{Cvoid}, devptr(a)::Ptr{Cvoid},
                                              Float32(slope)::Cfloat, Float32(scale)::Cfloat, devptr(bias)::Ptr{Cvoid},
                                              Cint(relu)::Cint, devptr(out)::Ptr{Cvoid}, C_NULL::Ptr{Cvoid},
                                              heads::Int64, (size(K, 1) ÷ heads)::Int64, stream_ptr()::Ptr{Cvoid})::Cint)
    return out
end

# ---- graph prep that the CUDA extension sends to the CPU (GNNGraphs/ext/GNNGraphsCUDAExt.jl:24-30) ---------------------------
function GNNGraphs.sort_edge_index(u::ROCVector{I}, v::ROCVector{I}) where {I <: Union{Int32, Int64}}
    uo, vo = similar(u), similar(v)
    check(@ccall libgnnmp.gnnmp_sort_edge_index(devptr(u)::Ptr{Cvoid}, devptr(v)::Ptr{Cvoid}, sizeof(I)::Cint, 1::Cint,
                                                length(u)::Int64, devptr(uo)::Ptr{Cvoid}, devptr(vo)::Ptr{Cvoid},
                                                stream_ptr()::Ptr{Cvoid})::Cint)
    return uo, vo
end

# ---- adjoints (needs ChainRulesCore as a further weakdep) ---------------------------------------------------------------
# Zygote reaches the methods above only if they carry rrules.  The pullback of the fused propagate w.r.t. xj is the same
# kernel on the plan of the reversed edge index; w.r.t. the edge weights it is one dot product per edge; max / min have
# their own kernel.  (Mirror of graphneuralnetworks.jl_amd/gnnmp/backward.py, which is tested against NNlib's rules.)
#
# using ChainRulesCore
# plan_t(g; self_loops = false) = (s, t = edge_index(g); Plan(t, s, g.num_nodes, g.num_nodes, self_loops))   # cached like plan()
#
# function ChainRulesCore.rrule(::typeof(fused_propagate), g, aggr::Union{typeof(+), typeof(mean)}, xj, w; kws...)
#     y = fused_propagate(g, aggr, xj, w; kws...)
#     function pullback(Δ)
#         Δ = unthunk(Δ); D = size(Δ, 1)
#         sd = aggr === mean ? 1f0 ./ max.(degree(g, Float32; dir = :in), 1f0) : nothing
#         Δx = similar(xj)
#         check(@ccall libgnnmp.gnnmp_propagate_f32(plan_t(g).handle::Ptr{Cvoid}, (w === nothing ? 0 : 1)::Cint, SUM::Cint,
#                   devptr(Δ)::Ptr{Cvoid}, devptr(w)::Ptr{Cvoid}, devptr(sd)::Ptr{Cvoid}, C_NULL::Ptr{Cvoid},
#                   devptr(Δx)::Ptr{Cvoid}, D::Int64, stream_ptr()::Ptr{Cvoid})::Cint)
#         Δw = w === nothing ? NoTangent() : begin
#             out = similar(w)
#             Δs = sd === nothing ? Δ : Δ .* sd'
#             check(@ccall libgnnmp.gnnmp_edge_dot_plan_f32(plan(g).handle::Ptr{Cvoid}, devptr(Δs)::Ptr{Cvoid},
#                       devptr(xj)::Ptr{Cvoid}, devptr(out)::Ptr{Cvoid}, D::Int64, stream_ptr()::Ptr{Cvoid})::Cint)
#             out
#         end
#         return NoTangent(), NoTangent(), NoTangent(), Δx, Δw
#     end
#     return y, pullback
# end
# (max / min: gnnmp_propagate_maxmin_grad_f32(plan_t(g).handle, xj, y, Δ, Δx, D, stream).)

end # module
