# bench/reference_cpu.jl — the reference ITSELF as the CPU baseline of bench.py (SURVEY.md §8d's preferred baseline, BASELINE.md §3.1).
#
#   julia --project=bench bench/reference_cpu.jl [--scale 32] [--threads-note]
#
# Times GCNConv(100 => 100, relu) and GATConv(100 => 16, heads = 8, relu) forward on a products-shaped synthetic graph at 1/scale of
# N = 2 449 029, E = 61 859 140 (the same degree law as graphneuralnetworks.jl_amd/gnnmp/synth.py: endpoints of rank
# floor(N u^1.79) under a random relabelling, E / 2 pairs listed in both directions, no self loops), CPU arrays, COO graph —
# the path GNNlib takes there: GCN through the SpMM fast path (GNNlib/src/msgpass.jl:215-238), GAT through gather -> message -> scatter.
# Protocol: BenchmarkTools.@benchmark like GraphNeuralNetworks/perf/perf.jl:25 (`res["CPU_FWD"] = @benchmark $m($g)`), median time.
# Prints ONE JSON line: {"value": edges/s of the two layers together, "unit": "edges/s", "cores": Threads.nthreads(),
# "blas_threads": ..., "sample": "...", "gcn_edges_per_s": ..., "gat_edges_per_s": ...}; bench.py copies it into `cpu_baseline`
# with kind = "reference".  bench.py runs this only when a `julia` is on PATH (there is none in the build image: never executed there).
using GraphNeuralNetworks, Flux, BenchmarkTools, Random, LinearAlgebra, Statistics

function parse_scale(args)
    i = findfirst(==("--scale"), args)
    return i === nothing ? 32 : parse(Int, args[i + 1])
end

function products_like(N::Int, E::Int; alpha = 1.79, seed = 2)
    @assert iseven(E)
    rng = MersenneTwister(seed)
    perm = randperm(rng, N)
    M = E ÷ 2
    draw() = perm[min.(floor.(Int, N .* rand(rng, M) .^ alpha), N - 1) .+ 1]
    u, v = draw(), draw()
    loop = u .== v
    v[loop] .= mod1.(v[loop] .+ 1, N)
    return vcat(u, v), vcat(v, u)
end

function main()
    scale = parse_scale(ARGS)
    N = 2_449_029 ÷ scale
    E = (61_859_140 ÷ scale) & ~1
    D, H, C = 100, 8, 16
    s, t = products_like(N, E)
    x = randn(MersenneTwister(1), Float32, D, N)
    g = GNNGraph((s, t); num_nodes = N, graph_type = :coo)
    gcn = GCNConv(D => D, relu)                       # add_self_loops = true: E' = E + N edges traversed
    gat = GATConv(D => C, relu; heads = H)
    Ep = E + N
    bg = @benchmark $gcn($g, $x)
    ba = @benchmark $gat($g, $x)
    tg, ta = median(bg).time * 1e-9, median(ba).time * 1e-9
    sample = "reference (GraphNeuralNetworks.jl) on CPU arrays, 1/$(scale)-scale products-shaped graph N=$N E'=$Ep D=$D: " *
             "GCNConv($D=>$D,relu) $(round(tg, digits = 3))s + GATConv($D=>$C,h=$H,relu) $(round(ta, digits = 3))s, BenchmarkTools median"
    println("{\"value\": $(2 * Ep / (tg + ta)), \"unit\": \"edges/s\", \"cores\": $(Threads.nthreads()), " *
            "\"blas_threads\": $(BLAS.get_num_threads()), \"sample\": \"$sample\", " *
            "\"gcn_edges_per_s\": $(Ep / tg), \"gat_edges_per_s\": $(Ep / ta)}")
end

main()
