# Hmc.jl -- Julia host module over libhmcgibbs.so (C ABI: include/hmcg.h).
#
# Drop-in for the data-parallel hot path of joe5saia/Hmc.jl: the same unexported names the
# reference's drivers call (code/run_hmm.jl:77,95,119,120) -- Hmc.estopt, Hmc.estimatemodel,
# Hmc.saveresults, Hmc.startdate/enddate/makey/yobs, Hmc.forecast, makedate -- plus the additive
# batched entry Hmc.estimatewindows.  All sampling happens in hand-written HIP kernels behind a
# thin ccall; this file is plumbing (argument marshalling, Julia-layout result arrays, CSV).
#
# STATUS: written against Julia >= 1.6 and NOT EXECUTED -- neither the build image nor the GPU box
# has a `julia` binary.  The tested host layer with the same surface is hmc.jl_amd/hmc.py; the C ABI
# it shares with this file is what the parity tests exercise.
#
# Wired: estimatemodel, estimatewindows (batched; `devices = [0,1,...]` partitions the windows over several GPUs through
# hmcg_estimate_batch_multi; `window_ids` pins the RNG streams), estimatesignals! (signal Monte-Carlo path incl. signals
# past the end date), saveresults with and without signals -- written by the library's native CSV writer
# (hmcg_save_results_csv: CSV.jl 0.5.16 float text, 250k rows x 5 files in under a second).  Not wired here (available
# through the C ABI and the Python host layer): checkpoint/resume, the smoothed-probability output, the correlation
# workbook writer (calccorr's matrices themselves: estimatewindows(...; corr=true)).  runaggregate /
# calcdispersion of the reference work unchanged on the files written here (same names, columns and float text).
#
# Reference lines mirrored: estopt src/Hmc.jl:17-73, accessors :85-107, makedate :573-582,
# forecast :658-667, basicsave/saveresults :707-748, estimatemodel :850-865.
module Hmc

using Dates
using Printf
using LinearAlgebra

export makedate

const LIBHMCG = get(ENV, "HMCG_LIBRARY", joinpath(@__DIR__, "..", "csrc", "libhmcgibbs.so"))
const HMCG_MAXH = 8

# ---- C structs (include/hmcg.h) ------------------------------------------------------------
struct hmcg_config
    struct_size::Int32
    W::Int32
    K::Int32
    ldY::Int32
    max_T::Int32
    burnin::Int32
    nrun::Int32
    H::Int32
    horizons::NTuple{HMCG_MAXH,Int32}
    seed::UInt64
    window_base::UInt32
    device::Int32
    flags::Int32
    threads_per_window::Int32
    sweep_base::Int32
    sweep_count::Int32
    alpha::Float64
    nu::Float64
    kappa::Float64
    n_samples::Int32
    blend_mask::Int32
    min_T::Int32                      # device entry: hint for the length-bucketed dispatch (host entries ignore it)
    reserved3::Int32
end

struct hmcg_extras
    struct_size::Int32
    reserved::Int32
    x_init::Ptr{Int32}
    x_final::Ptr{Int32}
    pif_final::Ptr{Float64}
    xstate::Ptr{UInt8}
    sumacc::Ptr{Float64}
    window_ids::Ptr{UInt32}
    sig_range::Ptr{Int32}
    save_range::Ptr{Int32}
    sigma_signal::Ptr{Float64}
    sigvals::Ptr{Float64}
    nsave_ld::Int32
    reserved2::Int32
    end_pos::Ptr{Int32}
    pi_smooth_mean::Ptr{Float64}
    pi_filter_mean::Ptr{Float64}
    corr::Ptr{Float64}
    pi_smooth_draws::Ptr{Float64}     # [W][K][ldY][nd] = (Nrun, N, D, W): samples.pib of every kept draw (src/Hmc.jl:552,558)
    sample_summary::Ptr{Float64}      # [W][n_samples][3K+K^2+2H]: runaggregate's (date, signalid) rows of a signal run (hmcg.h)
end
const HMCG_MAXTAIL = 256
const HMCG_MAXDEV = 16

struct hmcg_timing                    # include/hmcg.h (ABI 107)
    kernel_ms::Float64
    launches::Int32
    threads_per_window::Int32
    steps_per_thread::Int32
    lds_bytes::Int32
    helper_waves::Int32
    device::Int32
    call_ms::Float64
    windows::Int32
    occupancy::Int32
    buckets::Int32
    streaming::Int32
end

last_error() = unsafe_string(ccall((:hmcg_last_error, LIBHMCG), Cstring, ()))
device_count() = Int(ccall((:hmcg_device_count, LIBHMCG), Cint, ()))

# ---- options (field names and keyword defaults of the reference struct) ---------------------
mutable struct estopt
    rawdata::Vector{Float64}
    dates::Vector{Date}
    sampleRange::AbstractVector{Int}
    signalRange::AbstractVector{Int}
    signalSave::AbstractVector{Int}
    obsRange::AbstractVector{Int}
    endIndex::Int
    horizons::Vector{Int}
    D::Int
    burnin::Int
    Nrun::Int
    signalburnin::Int
    signalNrun::Int
    noise::Float64
    noiseSamples::Int
    σsignal::Float64
    series::String
    seed::Int
    function estopt(rawdata, dates; sampleRange=1:121, signalRange=2:1, signalSave=2:1, endIndex=121,
                    horizons=[12], D=3, burnin=1_000, Nrun=1_000, signalburnin=1_000, signalNrun=1_000,
                    noise=0.0, noiseSamples=1, σsignal=0.0, series="offical", seed=1234)
        issubset(signalRange, sampleRange) || @error "signalRange is not a subset of sampleRange"
        issubset(signalSave, signalRange) || @error "signalSave is not a subset of signalRange"
        new(Vector{Float64}(rawdata), Vector{Date}(dates), sampleRange, signalRange, signalSave,
            setdiff(sampleRange, signalRange), endIndex, collect(Int, horizons), D, burnin, Nrun,
            signalburnin, signalNrun, noise, noiseSamples, σsignal, series, seed)
    end
end

update_itators!(opt::estopt) = (opt.obsRange = setdiff(opt.sampleRange, opt.signalRange); nothing)
makey(x::estopt) = x.rawdata[x.sampleRange]
enddate(x::estopt, extra=0) = x.dates[x.endIndex + extra]
startdate(x::estopt) = x.dates[first(x.sampleRange)]
yobs(x::estopt, index) = x.rawdata[index]
yend(x::estopt, extra=0) = x.rawdata[x.endIndex + extra]

function makedate(x)
    y = Int(x)
    Dates.Date(y ÷ 12 + 1960, mod(y, 12) + 1)
end

function forecast(μ, A, πb, horizon, Yreal)
    f = dot(vec(πb' * A^horizon), μ)
    return f, f - Yreal
end

# ---- the ccall ---------------------------------------------------------------------------
function _check_live(opt::estopt; signals::Bool=false)
    (signals || isempty(opt.signalRange)) || error("this entry takes windows without signals; use estimatesignals!")
    (first(opt.sampleRange) == 1 && collect(opt.sampleRange) == collect(1:last(opt.sampleRange))) ||
        error("sampleRange must be 1:N (src/Hmc.jl indexes window-relative arrays with absolute indices)")
    if !isempty(opt.signalRange)
        sr = collect(opt.signalRange)
        (sr == collect(first(sr):last(sr)) && last(sr) == last(opt.sampleRange)) ||
            error("signalRange must be a contiguous tail of sampleRange")
        0 <= last(sr) - opt.endIndex <= HMCG_MAXTAIL || error("sigLen = last(signalRange) - endIndex must lie in 0..$(HMCG_MAXTAIL)")
    end
end

_yreal(opt::estopt) = [opt.endIndex + h <= length(opt.rawdata) ? opt.rawdata[opt.endIndex + h] : NaN for h in opt.horizons]

"""
    estimatewindows(opts::Vector{estopt}; device=0, devices=nothing, keepdraws=true, window_ids=nothing)

One call for many windows (all must share D, burnin, Nrun, horizons, seed) -- what the reference fans out as
`sbatch --array=120-579`, one Julia process per end date (slurmscripts/base_estimation.sh:5,17).  `devices = [0, 1, ..., 7]`
partitions the windows over those GPUs inside the library (hmcg_estimate_batch_multi: one host thread per device,
results gathered into these arrays; bit-identical to the single-device call).  `window_ids` (UInt32 per window) pins
the RNG streams.  Returns
`(samples::Vector{NamedTuple}, summary::Matrix{Float64}, status::Vector{Int32})`; `summary[:, w]` holds the
means of the 5-digit-rounded draws in the order mu | sigma | pib_end | A(:) | forecasts.  `corr=true` adds a fourth
value, `ρ[:, :, w]`: the correlation matrix calccorr (src/Hmc.jl:1094-1163) computes per end date from the per-draw CSV
files, accumulated on the device (extras.corr; labels: `corrnames`); with `keepdraws=false` no draw leaves the GPU.
"""
function estimatewindows(opts::Vector{estopt}; device::Integer=0, devices=nothing, keepdraws::Bool=true, window_ids=nothing,
                         corr::Bool=false)
    foreach(_check_live, opts)
    o = opts[1]
    W = length(opts); K = o.D; H = length(o.horizons); nrun = o.Nrun
    Ts = Int32[length(x.sampleRange) for x in opts]
    ldY = Int(maximum(Ts))
    Y = zeros(Float64, ldY, W)                       # column w = window w  (C: Y[w][t])
    yreal = zeros(Float64, max(H, 1), W)
    for (w, x) in enumerate(opts)
        Y[1:Ts[w], w] = makey(x)
        H > 0 && (yreal[1:H, w] = _yreal(x))
    end
    hz = ntuple(i -> i <= H ? Int32(o.horizons[i]) : Int32(0), HMCG_MAXH)
    cfg = Ref(hmcg_config(Int32(sizeof(hmcg_config)), W, K, ldY, Int32(maximum(Ts)), o.burnin, nrun, H, hz,
                          UInt64(o.seed), UInt32(0), Int32(device), Int32(0), Int32(0), Int32(0), Int32(0), 0.0, 0.0,
                          0.0, Int32(0), Int32(0), Int32(0), Int32(0)))
    NS = 3K + K * K + 2H
    μ = keepdraws ? Array{Float64}(undef, nrun, K, W) : Float64[]
    σ = keepdraws ? Array{Float64}(undef, nrun, K, W) : Float64[]
    A = keepdraws ? Array{Float64}(undef, nrun, K, K, W) : Float64[]
    πe = keepdraws ? Array{Float64}(undef, nrun, K, W) : Float64[]
    fc = keepdraws ? Array{Float64}(undef, nrun, 2H, W) : Float64[]
    summary = Array{Float64}(undef, NS, W)
    status = zeros(Int32, W)
    NC = 3K + K * K + 1
    ρ = corr ? Array{Float64}(undef, NC, NC, W) : Float64[]      # symmetric: C row-major == Julia column-major
    p(a) = isempty(a) ? Ptr{Float64}(C_NULL) : pointer(a)
    wids = window_ids === nothing ? UInt32[] : Vector{UInt32}(window_ids)
    ex = Ref(hmcg_extras(Int32(sizeof(hmcg_extras)), Int32(0), C_NULL, C_NULL, C_NULL, C_NULL, C_NULL,
                         isempty(wids) ? Ptr{UInt32}(C_NULL) : pointer(wids), C_NULL, C_NULL, C_NULL, C_NULL, Int32(0), Int32(0),
                         C_NULL, C_NULL, C_NULL, corr ? pointer(ρ) : Ptr{Float64}(C_NULL), C_NULL, C_NULL))
    yr = H > 0 ? pointer(yreal) : Ptr{Float64}(C_NULL)
    rc = GC.@preserve Y Ts yreal μ σ A πe fc summary status wids ρ begin
        if devices === nothing
            ccall((:hmcg_estimate_batch, LIBHMCG), Cint,
                  (Ref{hmcg_config}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ref{hmcg_extras}, Ptr{Cvoid}),
                  cfg, Y, Ts, yr, p(μ), p(σ), p(A), p(πe), p(fc), summary, status, ex, C_NULL)
        else
            devs = Vector{Int32}(devices)
            ccall((:hmcg_estimate_batch_multi, LIBHMCG), Cint,
                  (Ref{hmcg_config}, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                   Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ref{hmcg_extras}, Ptr{Cvoid}),
                  cfg, Int32(length(devs)), devs, Y, Ts, yr, p(μ), p(σ), p(A), p(πe), p(fc), summary, status, ex, C_NULL)
        end
    end
    rc == 0 || error("libhmcgibbs rc=$rc: $(last_error())")
    samples = NamedTuple[]
    if keepdraws
        for (w, x) in enumerate(opts)
            push!(samples, (μ = μ[:, :, w], σ = σ[:, :, w],
                            πb = reshape(πe[:, :, w], nrun, 1, K),   # only the last time step is produced (src/Hmc.jl:744,861)
                            A = A[:, :, :, w], forecasts = fc[:, :, w], obsdates = fill(enddate(x), nrun)))
        end
    end
    return corr ? (samples, summary, status, ρ) : (samples, summary, status)
end

"""
    corrnames(K, horizons) -> Vector{String}

Row / column labels of the matrices `estimatewindows(...; corr=true)` returns: what calccorr (src/Hmc.jl:1094-1163)
derives from the CSV headers -- μ1..K, σ1..K, π1..K, trans_i_j (i fastest), forecast_<first horizon>.
"""
corrnames(K::Integer, horizons) = vcat(["μ$i" for i in 1:K], ["σ$i" for i in 1:K], ["π$i" for i in 1:K],
                                       vec(["trans_$(i)_$(j)" for i in 1:K, j in 1:K]), ["forecast_$(horizons[1])"])

"""
    estimatemodel(opt) -> (μ, σ, πb, A, forecasts, obsdates)      (src/Hmc.jl:850-865)

`πb` has size (Nrun, 1, D): only `πb[:, end, :]` exists, which is all `saveresults` and `forecast` read.
"""
estimatemodel(opt::estopt; device::Integer=0) = estimatewindows([opt]; device=device)[1][1]

# One window on the signal path: n_samples chained noise samples of (burnin + nrun) sweeps (src/Hmc.jl:889-912).
function _signal_call(opt::estopt, burnin, nrun, n_samples, σsignal, κ, α, ν, devh::Vector{Int}, blend::Integer, endpos; device::Integer=0)
    Y = makey(opt); T = Int32[length(Y)]; K = opt.D; H = length(opt.horizons); nd = n_samples * nrun
    hz = ntuple(i -> i <= H ? Int32(devh[i]) : Int32(0), HMCG_MAXH)
    cfg = Ref(hmcg_config(Int32(sizeof(hmcg_config)), 1, K, length(Y), length(Y), burnin, nrun, H, hz, UInt64(opt.seed),
                          UInt32(0), Int32(device), Int32(0), Int32(0), Int32(0), Int32(0), Float64(α), Float64(ν), Float64(κ),
                          Int32(n_samples), Int32(blend), Int32(0), Int32(0)))
    sig = Int32[first(opt.signalRange) - 1, last(opt.signalRange)]                 # 0-based [begin, end)
    nsave = length(opt.signalSave)
    sv = nsave > 0 ? Int32[first(opt.signalSave) - 1, last(opt.signalSave)] : Int32[0, 0]
    ssig = Float64[σsignal]
    sigvals = zeros(Float64, max(nsave, 1), n_samples)                             # C: [n_samples][nsave_ld]
    ep = Int32[endpos]
    yreal = _yreal(opt)
    μ = Array{Float64}(undef, nd, K); σ = similar(μ); πe = similar(μ)
    A = Array{Float64}(undef, nd, K, K); fc = Array{Float64}(undef, nd, 2H); st = zeros(Int32, 1)
    rc = GC.@preserve Y T yreal μ σ A πe fc st sig sv ssig sigvals ep begin
        ex = Ref(hmcg_extras(Int32(sizeof(hmcg_extras)), Int32(0), C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL,
                             pointer(sig), pointer(sv), pointer(ssig), pointer(sigvals), Int32(max(nsave, 1)), Int32(0),
                             endpos >= 0 ? pointer(ep) : Ptr{Int32}(C_NULL), C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
        ccall((:hmcg_estimate_batch, LIBHMCG), Cint,
              (Ref{hmcg_config}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
               Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ref{hmcg_extras}, Ptr{Cvoid}),
              cfg, Y, T, H > 0 ? pointer(yreal) : Ptr{Float64}(C_NULL), μ, σ, A, πe, fc, C_NULL, st, ex, C_NULL)
    end
    rc == 0 || error("libhmcgibbs rc=$rc: $(last_error())")
    return μ, σ, πe, A, fc, sigvals
end

"""
    estimatesignals!(opt) -> (μ, σ, πb, A, forecasts, obsdates, signalvals, signalids)      (src/Hmc.jl:868-914)

Sets `opt.σsignal` from a base run when it is 0 (`:869-872`; upstream runs that one with HyperParams(Y, D): κ = 1,
α = ν = 1).  `πb` is (Ndraws, D) as upstream (`:900`).  Noise comes from the library's counter-based RNG.
"""
function estimatesignals!(opt::estopt; device::Integer=0)
    _check_live(opt; signals=true)
    isempty(opt.signalRange) && error("estimatesignals! needs a signalRange")
    H = length(opt.horizons)
    if isapprox(opt.σsignal, 0)
        _, σb, _, _, _, _ = _signal_call(opt, opt.burnin, opt.Nrun, 1, 0.0, 1.0, 1.0, 1.0, collect(Int, opt.horizons), 0, -1; device=device)
        opt.σsignal = sum(σb) / length(σb) * opt.noise
    end
    sigLen = last(opt.signalRange) - opt.endIndex                                   # :888
    devh = [h > sigLen ? h - sigLen : 0 for h in opt.horizons]                     # :907
    blend = sigLen > 0 ? sum(Int[(1 << (k - 1)) for (k, h) in enumerate(opt.horizons) if h == sigLen]) : 0
    μ, σ, πe, A, fc, sigvals = _signal_call(opt, opt.signalburnin, opt.signalNrun, opt.noiseSamples, opt.σsignal, opt.noise,
                                            2.0, 2.0, devh, blend, sigLen > 0 ? opt.endIndex - 1 : -1; device=device)
    for (k, h) in enumerate(opt.horizons)
        h < sigLen && (fc[:, 2k-1:2k] .= NaN)                                       # never assigned upstream (:906-910)
    end
    n = opt.signalNrun; nd = n * opt.noiseSamples
    nsave = length(opt.signalSave)
    signalvals = Array{Float64}(undef, nd, nsave)
    signalids = Array{Int64}(undef, nd)
    for s in 1:opt.noiseSamples
        r = n*(s-1)+1:n*s
        signalids[r] .= s                                                            # :903
        for j in 1:nsave
            signalvals[r, j] .= sigvals[j, s]                                        # :904
        end
    end
    return (μ = μ, σ = σ, πb = πe, A = A, forecasts = fc, obsdates = fill(enddate(opt), nd),
            signalvals = signalvals, signalids = signalids)
end

# ---- CSV output (layout of src/Hmc.jl:707-748) ---------------------------------------------
# CSV.jl 0.5.16 float text (shortest round-trip digits; integral values without a fraction; |x| < 1e-4 as
# <integer mantissa>e-<n>, e.g. 24e-11), produced by the library so that every host language prints the same bytes
function _fmt(x::Float64)
    buf = Vector{UInt8}(undef, 48)
    n = ccall((:hmcg_format_float, LIBHMCG), Cint, (Float64, Ptr{UInt8}), x, buf)
    return String(buf[1:n])
end

# The five per-window files of saveresults (:741-746) for draw arrays in Julia layout, through the library's writer
# (hmcg_save_results_csv).  μ, σ, πe: (n, K); A: (n, K, K); fc: (n, 2H); sigvals: (nsave, noiseSamples) or nothing.
function _save_native(dir, date, K, horizons, μ, σ, πe, A, fc; sigvals=nothing)
    mkpath(dir)
    H = length(horizons); n = size(μ, 1)
    hz = Vector{Int32}(horizons)
    dates = [string(date)]
    sv = sigvals === nothing ? Ptr{Float64}(C_NULL) : pointer(sigvals)
    nsmp = sigvals === nothing ? 0 : size(sigvals, 2); nsave = sigvals === nothing ? 0 : size(sigvals, 1)
    rc = GC.@preserve μ σ πe A fc hz dates sigvals begin
        ccall((:hmcg_save_results_csv, LIBHMCG), Cint,
              (Cstring, Int32, Ptr{Cstring}, Int32, Int32, Ptr{Int32}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
               Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Int32, Int32, Int32),
              dir, Int32(1), dates, Int32(K), Int32(H), hz, Int64(n), μ, σ, πe, A, H > 0 ? pointer(fc) : Ptr{Float64}(C_NULL),
              sv, Int32(nsmp), Int32(nsave), Int32(nsave), Int32(0), Int32(1))
    end
    rc == 0 || error("hmcg_save_results_csv rc=$rc (directory $dir)")
end

function basicsave(data, dates, fname, dataheader; precision=5, signal=Array{Float64}(undef, 0, 0), signalids=Int64[])
    hassig = size(signal, 2) > 0
    header = vcat(["date"], String.(dataheader))
    hassig && append!(header, ["signal_$i" for i in 1:size(signal, 2)])
    !isempty(signalids) && insert!(header, 2, "signalid")                          # :717
    open(fname, "w") do io
        println(io, join(header, ","))
        for i in 1:size(data, 1)
            cells = [string(dates[i])]
            hassig && push!(cells, string(signalids[i]))                           # only together with signals (:715-717)
            append!(cells, [_fmt(round(Float64(v); digits=precision)) for v in data[i, :]])
            hassig && append!(cells, [_fmt(round(Float64(v); digits=5)) for v in signal[i, :]])
            println(io, join(cells, ","))
        end
    end
end

function _savesignalresults(samples, opt, dir)
    h1 = ["state_$i" for i in 1:opt.D]
    h2 = vec(["trans_$(i)_$(j)" for i in 1:opt.D, j in 1:opt.D])
    h3 = String[]
    for h in opt.horizons
        push!(h3, "forecast_$h"); push!(h3, "forecast_error_$h")
    end
    mkpath(dir)
    kw = (signal = samples.signalvals, signalids = samples.signalids)
    n = size(samples.μ, 1)
    basicsave(samples.μ, samples.obsdates, joinpath(dir, "filtered_means_$(enddate(opt)).csv"), h1; kw...)
    basicsave(samples.σ, samples.obsdates, joinpath(dir, "filtered_variances_$(enddate(opt)).csv"), h1; kw...)
    basicsave(samples.πb, samples.obsdates, joinpath(dir, "filtered_state_probs_$(enddate(opt)).csv"), h1; kw...)
    basicsave(reshape(samples.A, n, :), samples.obsdates, joinpath(dir, "filtered_trans_probs_$(enddate(opt)).csv"), h2; kw...)
    basicsave(samples.forecasts, samples.obsdates, joinpath(dir, "forecasts_$(enddate(opt)).csv"), h3; kw...)
end

function saveresults(samples, opt, dir; hassignals=false)
    hassignals && return _savesignalresults(samples, opt, dir)                     # :735-739 (writes under `dir`)
    odir = "data/output/$(opt.series)/"          # the reference ignores `dir` here (src/Hmc.jl:741)
    _save_native(odir, enddate(opt), opt.D, opt.horizons, samples.μ, samples.σ, samples.πb[:, end, :], samples.A,
                 samples.forecasts)
end

end # module
