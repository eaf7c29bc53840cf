// Constants of the persistent LSTM the host side shares with the kernels (lstm_persist.h): plain data, no kernel definitions.
#pragma once

namespace ac {

constexpr int LP_D = 512, LP_SLICES = 32, LP_FLAG_STRIDE = 32;   // one 128-byte line per flag

// Sticky status words of a handle (host-pinned, device-mapped: the host reads them without synchronising; they are
// never cleared by a launch).  Written by lstm_tail_kernel / rvq_decode_kernel.
enum { ST_LSTM_TIMEOUT = 0, ST_LSTM_PLACEMENT = 1, ST_BAD_TOKEN = 2, ST_NONFINITE_CLIPS = 3, ST_WORDS = 4 };

}  // namespace ac
