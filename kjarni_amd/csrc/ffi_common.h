// Shared plumbing of the extern "C" layer: thread-local last error
// (crates/kjarni-ffi/src/error.rs:7-100) and exception -> error-code mapping.
// `panic = "abort"` in the reference means nothing unwinds across the ABI; here
// every entry point catches everything.
#pragma once
#include <exception>
#include <string>

#include "../../include/kjarni.h"
#include "encoder.h"

#define KJARNI_EXPORT extern "C" __attribute__((visibility("default")))

namespace kjarni {

void set_last_error(const std::string& msg);

struct ModelNotFound : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct InvalidConfig : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Runs fn(); maps exceptions to codes, recording the message.  `fallback` is
// the code for a generic failure (LoadFailed for *_new, InferenceFailed for run).
template <class F>
KjarniErrorCode guarded(KjarniErrorCode fallback, F&& fn)
{
    try {
        fn();
        return KJARNI_OK;
    } catch (const GpuUnavailable& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_GPU_UNAVAILABLE;
    } catch (const ModelNotFound& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_MODEL_NOT_FOUND;
    } catch (const InvalidConfig& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INVALID_CONFIG;
    } catch (const InvalidDeviceList& e) {
        set_last_error(e.what());
        return KJARNI_ERROR_INVALID_CONFIG;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return fallback;
    } catch (...) {
        set_last_error("unknown error");
        return KJARNI_ERROR_UNKNOWN;
    }
}

}  // namespace kjarni
