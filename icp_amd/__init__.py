"""icp_amd -- MI355X-native implementation of the drawpixels/icp hot path.

Sub-modules keep the reference's file and function names:
  icp_3d, icp_row, icp            rigid ICP
  nicp_3d, nicp_row, nicp_row_rt  non-rigid ICP
All numerics run in libicp_hip.so (HIP, gfx950) through the C-ABI declared in
include/icp_hip.h; importing the package does not need a GPU, calling it does.
"""
__version__ = "0.1.0"
