// cub/cub.cuh -- empty stand-in (TEST INFRASTRUCTURE, see oracle/oracle.h).  B/kernel_opt_geometry.cu and
// B/kernel_surfel_activation.cu include CUB but use nothing of it (their kernels have one thread per surfel and no block
// collective); the kernels that do (pose accumulation, PCG: B/gauss_newton.cuh) are not compiled for the host.
#pragma once
