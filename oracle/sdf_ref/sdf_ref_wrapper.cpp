// TEST INFRASTRUCTURE (oracle/_ref/libsdf_ref.so): the REFERENCE's SDF voxeliser, compiled for the host from its own source
// file where it lies (REF_SDF_CU = /root/reference/pose_data_optimize/sdf/sdf/csrc/sdf_cuda_kernel.cu, included below; nothing
// of it is copied into this repository), executed on the CPU by the fiber-based HIP shim of tests/hipcpu.  Pins
// oracle/sdf_oracle.py and generates tests/golden/sdf_ref.npz (tests/golden/make_sdf_golden.py).
// Launch geometry as the reference's host wrapper (sdf_cuda_kernel.cu:311-319): 512 threads, blocks = voxels / 512 ROUNDED
// DOWN -- voxels beyond blocks * 512 are never written (the caller's initial phi stays).
#include REF_SDF_CU

extern "C" int sdf_ref_f32(float* phi, const int32_t* faces, const float* vertices, int batch_size, int num_faces,
                           int num_vertices, int grid_size) {
    const long total = (long)batch_size * grid_size * grid_size * grid_size;
    const int threads = 512;
    const unsigned blocks = (unsigned)(total / threads);
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(sdf_cuda_kernel<float>, dim3(blocks), dim3(threads), 0, nullptr, phi, faces, vertices, batch_size,
                       num_faces, num_vertices, grid_size);
    return 0;
}
