// test/BenchmarkStencil.cpp of the reference as a plain C++ program above the lsk C ABI: same flags
// (-dim -nx -ny -nz -vp -solver -it -pt), same set-up (two identical systems sol/rhs and sol1/rhs1
// on one matrix, rhs = 1, x0 = 0), same warm-up / timed trace structure (:201-262) and the same
// three output lines.  Extensions for tests: -systems 1 solves a single system, -coo 1 builds the
// COO form of the matrix, -f32 1 uses float entries, -restart N sets GMRES's restart length
// (the reference hard-codes 10), -history 1 prints the solver's scalar history one value per line,
// -compress 0 / -sliced 0 / -fuse 0 turn the 32-bit structure, the sliced-pattern copy and
// the held-back updates off, -redraws N re-makes the set-up-time copies N times keeping the fastest (off by default), -adjoint N (after the solve) multiplies the solution by the
// transpose and prints it against the product itself (the stencil matrices are symmetric) and both
// sides of <A x, b> = <x, A^T b>; N > 1 transposes in runs of N entries; -binned 1 (with -coo 1) multiplies
// through the binned copy of the COO matrix, in blocks of -bincols columns and tiles of -binrows rows
// (0: the library's choice); -cgs 1 gives GMRES the classical Gram-Schmidt option; -pair 0 multiplies the two
// systems one after the other instead of in one pass over the matrix; -uniform 1 sets the library's option
// "sliced.uniform" (a constant-coefficient stencil's values as scalars: the products read no value array);
// -graph 1 makes the traces real: the first warm-up trace runs as it is, then the -pt steps of a trace are
// recorded once as a HIP graph (LegionSolversHost::Trace) and every further trace is one replay (the
// reference records and replays Legion traces here, :219-222; -history then prints the last values only).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "stencil_generator.hpp"

using namespace LegionSolversHost;

namespace {

struct Args {
    long dim = 1, solver = 1, nx = 16, ny = 16, nz = 16, vp = 4, it = 20, pt = 1;
    long systems = 2, coo = 0, f32 = 0, restart = 10, history = 0, compress = 1, fuse = 1, redraws = 0, adjoint = 0, sliced = 1;
    long binned = 0, bincols = 0, binrows = 0, cgs = 0, pair = 1, uniform = 0, graph = 0;
};

Args parse(int argc, char **argv) {
    Args a;
    const struct { const char *flag; long Args::*field; } table[] = {
        {"-dim", &Args::dim}, {"-solver", &Args::solver}, {"-nx", &Args::nx}, {"-ny", &Args::ny},
        {"-nz", &Args::nz}, {"-vp", &Args::vp}, {"-it", &Args::it}, {"-pt", &Args::pt},
        {"-systems", &Args::systems}, {"-coo", &Args::coo}, {"-f32", &Args::f32}, {"-restart", &Args::restart},
        {"-history", &Args::history}, {"-compress", &Args::compress}, {"-fuse", &Args::fuse},
        {"-redraws", &Args::redraws}, {"-adjoint", &Args::adjoint},
        {"-sliced", &Args::sliced}, {"-binned", &Args::binned}, {"-bincols", &Args::bincols},
        {"-binrows", &Args::binrows}, {"-cgs", &Args::cgs}, {"-pair", &Args::pair},
        {"-uniform", &Args::uniform}, {"-graph", &Args::graph}};
    for (int i = 1; i + 1 < argc; i += 2) {
        bool known = false;
        for (const auto &t : table)
            if (!std::strcmp(argv[i], t.flag)) {
                a.*(t.field) = std::atol(argv[i + 1]);
                known = true;
            }
        if (!known) throw std::invalid_argument(std::string("unknown flag ") + argv[i]);
    }
    return a;
}

template <typename Solver> void print_history(const Solver &) {}
template <typename T> void print_history(const CGSolver<T> &s) {
    for (const auto &v : s.residual_norm_squared) std::printf("%.17g\n", static_cast<double>(v.get()));
}
template <typename T> void print_history(const BiCGStabSolver<T> &s) {
    for (const auto &v : s.rho) std::printf("%.17g\n", static_cast<double>(v.get()));
}

template <typename T, typename Solver>
void run(const Context &ctx, const Args &a, Solver &solver, SquarePlanner<T> &planner) {
    const long num_warmup_traces = 5;
    const long num_traces = (a.it + a.pt - 1) / a.pt;
    ctx.synchronize();
    std::unique_ptr<Trace<T, Solver>> trace;
    auto one_trace = [&](long i) {
        if (a.graph && i > 0) {
            if (!trace) trace = std::make_unique<Trace<T, Solver>>(ctx, planner, solver, a.pt);
            trace->replay();
        } else {
            for (long j = 0; j < a.pt; ++j) solver.step();
        }
    };
    for (long i = 0; i < num_warmup_traces; ++i) one_trace(i);
    ctx.synchronize();
    if (!planner.workspace_product_ms().empty()) { // (where the planner put its workspace, and by what timings)
        std::fprintf(stderr, "[LegionSolversHost] workspace vectors behind %d block(s) of a third of the device; product by blocks in front:",
                     planner.workspace_blocks_in_front());
        for (const float ms : planner.workspace_product_ms()) std::fprintf(stderr, " %.3f", ms);
        std::fprintf(stderr, " ms\n");
    }
    const auto begin = std::chrono::steady_clock::now();
    std::cout << "Performed " << num_warmup_traces << " warmup traces." << std::endl;
    for (long i = num_warmup_traces; i < num_traces; ++i) one_trace(i);
    ctx.synchronize();
    const auto end = std::chrono::steady_clock::now();
    std::cout << "Performed " << (num_traces - num_warmup_traces) << " timed traces." << std::endl;
    const double ns = std::chrono::duration<double, std::nano>(end - begin).count();
    const double ns_per_iter = ns / static_cast<double>(a.pt * std::max<long>(num_traces - num_warmup_traces, 1));
    std::cout << "Achieved " << ns_per_iter / 1.0e6 << "ms per iteration." << std::endl;
    if (a.graph && trace) std::printf("trace_nodes %lld\n", static_cast<long long>(trace->nodes()));
    if (a.history) {
        print_history(solver);
        const auto x = planner.get_vector(0, 0).to_host();
        double sum = 0.0, sum2 = 0.0;
        for (const T v : x) {
            sum += static_cast<double>(v);
            sum2 += static_cast<double>(v) * static_cast<double>(v);
        }
        std::printf("solution_sum %.17g\nsolution_sum_of_squares %.17g\n", sum, sum2);
        std::printf("paired_products %ld\n", planner.paired_products());
    }
    if (a.adjoint) { // vector 2 is workspace of every solver and free once the steps are done
        planner.rmatvec(2, 0);
        const auto atx = planner.get_vector(2, 0).to_host();
        planner.matvec(2, 0);
        const auto ax = planner.get_vector(2, 0).to_host();
        double diff = 0.0, scale = 0.0;
        for (size_t i = 0; i < ax.size(); ++i) {
            diff = std::max(diff, std::fabs(static_cast<double>(atx[i]) - static_cast<double>(ax[i])));
            scale = std::max(scale, std::fabs(static_cast<double>(ax[i])));
        }
        const double ax_b = static_cast<double>(planner.dot(2, 1).get());
        planner.rmatvec(2, 1);
        const double x_atb = static_cast<double>(planner.dot(0, 2).get());
        std::printf("transposed_product_max_difference %.3g of %.3g\nadjoint %.17g %.17g\n", diff, scale, ax_b, x_atb);
    }
}

template <typename T> int main_typed(const Args &a) {
    Context ctx(0);
    const Stencil st = benchmark_stencil(static_cast<int>(a.dim));
    std::vector<int64_t> extent = {a.nx};
    if (st.dim >= 2) extent.push_back(a.ny);
    if (st.dim >= 3) extent.push_back(a.nz);
    int64_t rows = 1;
    for (int64_t e : extent) rows *= e;
    const EqualPartition partition(rows, static_cast<int>(a.vp));

    std::unique_ptr<AbstractMatrix<T>> matrix;
    if (a.coo) {
        auto coo = std::make_unique<COOMatrix<T>>(create_linearized_coo_stencil_matrix<T>(ctx, extent, st, partition));
        if (a.binned && !coo->bin(static_cast<int>(a.bincols), static_cast<int>(a.binrows)))
            throw std::runtime_error("the binned copy was declined");
        matrix = std::move(coo);
    } else {
        // (constructed in place: a CSRMatrix is registered by address as an owner of optional memory and does not move)
        std::unique_ptr<CSRMatrix<T>> csr(new CSRMatrix<T>(create_linearized_csr_stencil_matrix<T>(
            ctx, extent, st, partition, a.compress != 0, static_cast<int>(a.redraws), a.sliced != 0)));
        if (a.adjoint > 1) csr->transpose_chunk = a.adjoint;
        matrix = std::move(csr);
    }

    std::vector<std::unique_ptr<PartitionedVector<T>>> sols, rhss;
    SquarePlanner<T> planner(ctx);
    planner.set_fusion(a.fuse != 0);
    planner.set_pairing(a.pair != 0);
    check(lsk_set_option("sliced.uniform", a.uniform != 0));
    for (long s = 0; s < a.systems; ++s) {
        sols.push_back(std::make_unique<PartitionedVector<T>>(ctx, s ? "sol1" : "sol", partition));
        sols.back()->zero_fill();
        planner.add_sol_vector(*sols.back());
    }
    for (long s = 0; s < a.systems; ++s) {
        rhss.push_back(std::make_unique<PartitionedVector<T>>(ctx, s ? "rhs1" : "rhs", partition));
        rhss.back()->constant_fill(T(1));
        planner.add_rhs_vector(*rhss.back());
    }
    for (long s = 0; s < a.systems; ++s) planner.add_row_partitioned_matrix(*matrix, static_cast<int>(s), static_cast<int>(s));

    if (a.solver == 1) {
        CGSolver<T> solver(planner);
        run<T>(ctx, a, solver, planner);
    } else if (a.solver == 2) {
        BiCGStabSolver<T> solver(planner);
        run<T>(ctx, a, solver, planner);
    } else if (a.solver == 3) {
        GMRESSolver<T> solver(planner, static_cast<int>(a.restart), a.cgs != 0);
        run<T>(ctx, a, solver, planner);
    } else {
        std::cout << "INVALID SOLVER" << std::endl;
        return 1;
    }
    return 0;
}

} // namespace

int main(int argc, char **argv) {
    try {
        const Args a = parse(argc, argv);
        return a.f32 ? main_typed<float>(a) : main_typed<double>(a);
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
