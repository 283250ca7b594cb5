// cub/device/device_scan.cuh -- empty stand-in (TEST INFRASTRUCTURE, see oracle/oracle.h): included by B/kernel_assign_colors.cu and
// B/kernel_delete_surfels.cu, which use no device-wide scan.
#pragma once
