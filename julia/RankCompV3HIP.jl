# Drop-in replacement for `identify_degs` of RankCompV3.jl (src/RankCompV3.jl:339-438) that runs the
# pair loop, the tallies and the iteration on an MI355X through libreo_hip.so (include/reo_hip.h).
# Same positional signature, same return value (G x (1 + 16 C) Matrix{Any}; C = 1 for two groups).  To use it inside
# the package: `include("RankCompV3HIP.jl")` after the original definition, or replace the call at
# src/RankCompV3.jl:652 with `RankCompV3HIP.identify_degs(...)`.
# NOT EXECUTED in this repository's pipeline (no Julia toolchain in the image); kept logic-free.
module RankCompV3HIP

const LIB = get(ENV, "LIBREO_HIP", "libreo_hip")

# The library keeps the device and pinned blocks of destroyed contexts in a process-wide cache (a context per call then costs nothing:
# include/reo_hip.h, reo_trim_memory; REO_DEVICE_CACHE_MB bounds it).  Other allocators of the process (AMDGPU.jl) cannot see that
# memory: `trim()` hands it back to the driver -- call it when the last identify_degs of a session is done; it also runs at exit.
trim() = (ccall((:reo_trim_memory, LIB), Int32, ()); nothing)
__init__() = atexit(trim)

function check(rc::Int32)
    rc == 0 && return
    msg = unsafe_string(ccall((:reo_last_error, LIB), Cstring, ()))
    rc == -1 ? throw(DimensionMismatch(msg)) : error("libreo_hip status $rc: $msg")
end

function identify_degs(data::AbstractMatrix, group::AbstractVector, gene_names::AbstractVector,
                       pval_reo::AbstractFloat, pval_deg::AbstractFloat, padj_deg::AbstractFloat,
                       ref_gene::BitVector, n_iter::Int64, n_conv::Int64;
                       seed::UInt64 = rand(UInt64), device::Integer = -1, n_gpus::Integer = 1)
    r, c = size(data)
    glev = unique(group)                                            # :353
    c == length(group) || throw(DimensionMismatch("'data' and 'group' do not have compatiable sizes"))
    length(glev) > 1   || throw(DimensionMismatch("Only 1 level in 'group1, at least 2 levels!"))
    gid = Int32[findfirst(==(g), glev) - 1 for g in group]          # 0-based, first-appearance order
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    if n_gpus == 1                                                  # one GPU; n_gpus = 0: every visible GPU (RCCL inside the library)
        check(ccall((:reo_create, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32, UInt64), ctx, device, seed))
    else
        check(ccall((:reo_create_multi, LIB), Int32, (Ref{Ptr{Cvoid}}, Int32, UInt64), ctx, n_gpus, seed))
    end
    try
        # groups and thresholds BEFORE the matrix: reo_set_matrix_* then ranks samples as their columns arrive and starts the
        # pair kernel's group-1 side while group 2 is still on its way over PCIe (include/reo_hip.h)
        check(ccall((:reo_set_groups, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Int64, Int32), ctx[], gid, c, length(glev)))
        check(ccall((:reo_compute_thresholds, LIB), Int32, (Ptr{Cvoid}, Float64), ctx[], pval_reo))           # :362
        if eltype(data) <: Integer
            X = convert(Matrix{Int64}, data)            # (no copy when `data` is a Matrix{Int64} already: Matrix(df_expr) of counts)
            check(ccall((:reo_set_matrix_i64, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Int64, Int64), ctx[], X, r, c, r))
        else
            X = convert(Matrix{Float64}, data)
            check(ccall((:reo_set_matrix_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64), ctx[], X, r, c, r))
        end
        res = Matrix{Any}(reshape(gene_names, r, 1))
        ref0 = UInt8.(ref_gene)
        for k in 1:length(glev)                                                                               # :396
            check(ccall((:reo_build_pairs, LIB), Int32, (Ptr{Cvoid}, Int32), ctx[], k - 1))                   # :363-392
            result = zeros(Float64, r, 15)                                                                    # :398
            iters = Ref{Int32}(0)
            trace = zeros(Int32, 2, max(n_iter, 1))
            check(ccall((:reo_identify_degs, LIB), Int32,
                        (Ptr{Cvoid}, Ptr{UInt8}, Float64, Float64, Int32, Int32, Ptr{Float64}, Ref{Int32}, Ptr{Int32}),
                        ctx[], ref0, pval_deg, padj_deg, n_iter, n_conv, result, iters, trace))                # :400-425
            for p in 1:iters[]
                @info "INFO: iteration $(p-1),  # DEGs $(trace[1,p]), # non-DEGs $(trace[2,p])"                # :418
            end
            gene_up_down = fill("no change", r)                                                               # :426-429
            sig = (result[:, 1] .<= pval_deg) .& (result[:, 2] .<= padj_deg)
            gene_up_down[sig .& (result[:, 15] .> 0)] .= "up"
            gene_up_down[sig .& (result[:, 15] .< 0)] .= "down"
            res = hcat(res, result, gene_up_down)                                                             # :430
            length(glev) == 2 && break                                                                        # :431-434
        end
        return res                                                                                            # :437
    finally
        ccall((:reo_destroy, LIB), Cvoid, (Ptr{Cvoid},), ctx[])
    end
end

end # module
