# The Legion_USE_HIP branch of LegionSolvers' CMakeLists.txt: what the reference does for
# Legion_USE_CUDA at CMakeLists.txt:35-43 (sources) and :53-60 (libraries), for a Legion built with
# HIP.  include() it from CMakeLists.txt after the LegionSolvers_SOURCES list is made and before
# add_library(LegionSolvers ...), with
#   LSK_ROOT   the checkout of this repository (include/lsk.h, legion_glue/, legionsolvers_amd/liblsk.so)
# NOT EXERCISED IN THIS REPOSITORY (no Legion here): a fragment for the maintainer, like the task
# bodies it adds.
#
# Beside it, three preprocessor guards admit LEGION_USE_HIP where they test LEGION_USE_CUDA
# (src/TaskBaseClasses.hpp:211,304, src/LegionUtilities.hpp:68, src/Initialize.cpp:14,484),
# src/CudaLibs.hpp:7-9,12 includes <cublas_v2.h> / <cuda_runtime.h> / <cusparse.h> and "CUDAUtilities.hpp"
# only under LEGION_USE_CUDA, and src/CudaLibs.cu:11-20 (the CUDA task body) likewise.

if(Legion_USE_HIP AND NOT Legion_USE_Kokkos)
    if(NOT DEFINED LSK_ROOT)
        message(FATAL_ERROR "LegionSolvers on HIP: set LSK_ROOT to the liblsk checkout")
    endif()
    # the HIP counterparts of src/CUDAUtilities.cpp, src/LinearAlgebraTasks.cu, src/COOMatrixTasks.cu and
    # src/CSRMatrixTasks.cu are ONE translation unit, plain C++ (no device code: the kernels are in
    # liblsk.so); src/CudaLibs.cu keeps its registration code (:23-66) and is compiled as C++; its task body
    # (:11-20) goes behind `#if !defined(LEGION_USE_HIP)`, lsk_legion_tasks.cpp defines the HIP one
    list(APPEND LegionSolvers_SOURCES
        ${LSK_ROOT}/legion_glue/lsk_legion_tasks.cpp
        src/CudaLibs.cu
    )
    set_source_files_properties(src/CudaLibs.cu PROPERTIES LANGUAGE CXX)
    find_package(hip REQUIRED)                       # hip::host: <hip/hip_runtime_api.h> for hipGetDevice
    find_library(LSK_LIBRARY lsk PATHS ${LSK_ROOT}/legionsolvers_amd NO_DEFAULT_PATH REQUIRED)
endif()

# ... add_library(LegionSolvers ${LegionSolvers_SOURCES}) and target_link_libraries(LegionSolvers Legion::Legion)
# stay as they are (CMakeLists.txt:46-48); then, in place of CUDA::cudart CUDA::cublas CUDA::cusparse (:55-59):

function(legion_solvers_link_lsk target)
    if(Legion_USE_HIP AND NOT Legion_USE_Kokkos)
        target_include_directories(${target} PRIVATE ${LSK_ROOT}/include ${LSK_ROOT}/legion_glue)
        target_link_libraries(${target} ${LSK_LIBRARY} hip::host)
        set_property(TARGET ${target} APPEND PROPERTY BUILD_RPATH ${LSK_ROOT}/legionsolvers_amd)
    endif()
endfunction()
