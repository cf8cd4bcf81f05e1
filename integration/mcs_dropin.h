// mcs_dropin.h — shared by the drop-in translation units (integration/*_mcs.cpp): ONE libmcs_hip context per process and the lock that serialises its use.
// Not part of the reference's headers and not installed; the reference's own headers stay unmodified.
#pragma once
#include <mutex>
#include "mcs_c.h"

namespace MultiColSLAM
{
namespace mcs_dropin
{
	mcs_ctx* context();        // created on first use (device 0, its own stream); throws std::runtime_error when there is no HIP device
	std::mutex& mutex();       // every call into the context's extractors happens under this lock (one stream serves them in turn)
	void check(int rc, const char* what);
}
}
