// What an application linked against LegionSolvers-on-liblsk may call besides LegionSolvers' own API.
// Part of the Legion-side binding (legion_glue/lsk_legion_tasks.cpp), not of liblsk.so.
#pragma once

namespace LegionSolvers {

// Only for applications that run with LSK_SLICED=1 (the CSR task body then multiplies stencil-like
// matrices through a set-up-time copy that holds the entry VALUES, which the reference's task reads
// from the region on every call, src/CSRMatrixTasks.cu:88-118): call this, from any thread, after
// rewriting the entries of the matrix whose column field instance starts at `col_ptr` on some GPU; the
// next product of every GPU processor that holds a copy made from that instance refills it from the
// region before it multiplies.  Without LSK_SLICED=1 every product reads live values and the call does
// nothing.
void lsk_legion_matrix_values_changed(const void *col_ptr);

} // namespace LegionSolvers
