// Stencil matrices of the reference's tests and benchmark (src/StencilGenerator.{hpp,cpp},
// src/ExampleSystems.cpp, test/BenchmarkStencil.cpp:33-131), generated on the device through the
// lsk set-up entry points; output identical to the reference's fill tasks with
// IndexOrder::ROW_MAJOR (offsets sorted lexicographically, rows numbered row-major).
#pragma once

#include <array>
#include <numeric>

#include "legion_solvers_host.hpp"

namespace LegionSolversHost {

struct Stencil {
    int dim = 1;
    std::vector<std::array<int64_t, 3>> offsets; // first `dim` components used
    std::vector<double> coeffs;
};

// The stencils of test/BenchmarkStencil.cpp's -dim flag: 1: 1-D [-1 2 -1]; 2: 2-D 5-point;
// 3: 3-D 7-point; 4: 3-D 27-point (centre 88/26, faces -6/26, edges -3/26, corners -2/26).
inline Stencil benchmark_stencil(int dim_flag) {
    Stencil s;
    if (dim_flag == 1) {
        s.dim = 1;
        s.offsets = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}};
        s.coeffs = {2.0, -1.0, -1.0};
    } else if (dim_flag == 2) {
        s.dim = 2;
        s.offsets = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}};
        s.coeffs = {4.0, -1.0, -1.0, -1.0, -1.0};
    } else if (dim_flag == 3) {
        s.dim = 3;
        s.offsets = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
        s.coeffs = {6.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0};
    } else if (dim_flag == 4) {
        s.dim = 3;
        const double by_weight[4] = {88.0 / 26.0, -6.0 / 26.0, -3.0 / 26.0, -2.0 / 26.0};
        for (int64_t a = -1; a <= 1; ++a)
            for (int64_t b = -1; b <= 1; ++b)
                for (int64_t c = -1; c <= 1; ++c) {
                    s.offsets.push_back({a, b, c});
                    s.coeffs.push_back(by_weight[std::abs(a) + std::abs(b) + std::abs(c)]);
                }
    } else {
        throw std::invalid_argument("INVALID DIM"); // test/BenchmarkStencil.cpp:132-135
    }
    return s;
}

namespace detail {

template <typename T> struct Generated {
    DeviceArray<int64_t> rects, row, col;
    DeviceArray<T> val;
    int64_t rows = 0, nnz = 0;
};

template <typename T>
Generated<T> generate(const Context &ctx, const std::vector<int64_t> &extent, Stencil st, bool coo) {
    if (static_cast<int>(extent.size()) != st.dim) throw std::invalid_argument("grid and stencil dimensions differ");
    // compare_row_major on the point, then the entry (src/StencilGenerator.cpp:40-52)
    std::vector<size_t> order(st.offsets.size());
    std::iota(order.begin(), order.end(), size_t(0));
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return std::make_pair(st.offsets[a], st.coeffs[a]) < std::make_pair(st.offsets[b], st.coeffs[b]);
    });
    std::vector<int64_t> offs;
    std::vector<T> cfs;
    for (size_t i : order) {
        for (int d = 0; d < st.dim; ++d) offs.push_back(st.offsets[i][d]);
        cfs.push_back(static_cast<T>(st.coeffs[i]));
    }
    Generated<T> g;
    g.rows = 1;
    for (int64_t e : extent) g.rows *= e;
    const int n_off = static_cast<int>(order.size());
    lsk_stream_t s = ctx.stream();
    DeviceArray<int64_t> counts(g.rows), prefix(g.rows);
    check(lsk_stencil_count(st.dim, extent.data(), n_off, offs.data(), 0, g.rows - 1, counts.get(), s));
    check(lsk_inclusive_sum_i64(g.rows, counts.get(), prefix.get(), s));
    g.nnz = g.rows ? prefix.to_host(ctx, g.rows - 1, 1)[0] : 0;
    g.rects = DeviceArray<int64_t>(2 * g.rows);
    check(lsk_rects_from_counts_i64(g.rows, counts.get(), prefix.get(), 0, g.rects.get(), s));
    g.col = DeviceArray<int64_t>(g.nnz);
    g.val = DeviceArray<T>(g.nnz);
    if (coo) {
        g.row = DeviceArray<int64_t>(g.nnz);
        check(Abi<T>::stencil_fill_coo(st.dim, extent.data(), n_off, offs.data(), cfs.data(), 0, g.rows - 1,
                                       g.rects.get(), 0, g.row.get(), g.col.get(), g.val.get(), s));
    } else {
        check(Abi<T>::stencil_fill_csr(st.dim, extent.data(), n_off, offs.data(), cfs.data(), 0, g.rows - 1,
                                       g.rects.get(), 0, g.col.get(), g.val.get(), s));
    }
    ctx.synchronize(); // the host argument arrays must outlive the launches
    return g;
}

} // namespace detail

// create_linearized_csr_stencil_matrix (src/StencilGenerator.hpp; fill task
// src/StencilGenerator.cpp:379-517) for the grid [0, extent), rows coloured by `partition`.
template <typename T>
CSRMatrix<T> create_linearized_csr_stencil_matrix(const Context &ctx, const std::vector<int64_t> &extent,
                                                  const Stencil &stencil, const EqualPartition &partition,
                                                  bool compress = true, int redraws = 4, bool slice = true) {
    auto g = detail::generate<T>(ctx, extent, stencil, false);
    if (partition.n != g.rows) throw std::invalid_argument("the partition must colour the linearised grid");
    return CSRMatrix<T>(ctx, partition, g.rows, std::move(g.rects), std::move(g.col), std::move(g.val), compress,
                        redraws, slice);
}

// create_linearized_coo_stencil_matrix (fill task src/StencilGenerator.cpp:129-242).
template <typename T>
COOMatrix<T> create_linearized_coo_stencil_matrix(const Context &ctx, const std::vector<int64_t> &extent,
                                                  const Stencil &stencil, const EqualPartition &partition) {
    auto g = detail::generate<T>(ctx, extent, stencil, true);
    if (partition.n != g.rows) throw std::invalid_argument("the partition must colour the linearised grid");
    std::vector<std::pair<int64_t, int64_t>> entries;
    for (int c = 0; c < partition.colors(); ++c) {
        if (partition.volume(c) <= 0) {
            entries.emplace_back(0, 0);
            continue;
        }
        const int64_t first = g.rects.to_host(ctx, 2 * partition.piece[c].first, 1)[0];
        const int64_t last = g.rects.to_host(ctx, 2 * partition.piece[c].second + 1, 1)[0];
        entries.emplace_back(first, last - first + 1);
    }
    return COOMatrix<T>(ctx, partition, g.rows, std::move(g.row), std::move(g.col), std::move(g.val),
                        std::move(entries));
}

} // namespace LegionSolversHost
