// (the host-side emulation of the column engine links no hipcub: tests/cpp/hipshim/hip/hip_runtime.h)
#pragma once
