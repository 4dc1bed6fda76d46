#!/bin/bash
# The host concurrency core of libembodied_hip.so under ThreadSanitizer and under
# AddressSanitizer + UndefinedBehaviorSanitizer: csrc/replay_abi.cpp, index_abi.cpp,
# kernels_abi.cpp (with replay_index.h, selectors.h, defer_gate.h, stream_order.h,
# device_rings.h) compiled by g++ against tests/fake_hip -- host memory as device
# memory, launches executed at once by plain loops -- and driven by
# tests/sanitize/soak.cpp through the C ABI: helper thread on, predicted rows,
# carried publish, four sampler threads, checkpoint-style bookkeeping, fork.
#
#   tools/run_sanitizers.sh [seconds per run, default 20] [output dir, default profiles]
#
# Writes <dir>/r06_tsan.txt and <dir>/r06_asan.txt; exit status 0 = no report.
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
SECONDS_EACH="${1:-20}"
OUT="${2:-$R/profiles}"
WORK="${TMPDIR:-/tmp}/emb_sanitize.$$"
mkdir -p "$WORK" "$OUT"
SRC="$R/embodied_amd/csrc/replay_abi.cpp $R/embodied_amd/csrc/index_abi.cpp $R/embodied_amd/csrc/kernels_abi.cpp \
     $R/tests/fake_hip/fake_kernels.cpp $R/tests/sanitize/soak.cpp"
FLAGS="-std=c++17 -O1 -g -fno-omit-frame-pointer -I$R/tests/fake_hip/include -I$R/embodied_amd/csrc -pthread"
status=0
build() {   # name, sanitizer flags
  g++ $FLAGS $2 $SRC -o "$WORK/soak_$1" 2> "$WORK/build_$1.log" || { cat "$WORK/build_$1.log"; exit 3; }
}
run() {     # report file, binary, args...
  local report="$1" bin="$2"; shift 2
  echo "\$ HIP_FORCE_DEV_KERNARG=$HIP_FORCE_DEV_KERNARG${EMB_DEFER_INDEX:+ EMB_DEFER_INDEX=$EMB_DEFER_INDEX}${EMB_PREDICT_ROWS:+ EMB_PREDICT_ROWS=$EMB_PREDICT_ROWS} $(basename "$bin") $*" >> "$report"
  "$bin" "$@" >> "$report" 2>&1
  local rc=$?
  echo "exit status $rc" >> "$report"
  [ $rc -ne 0 ] && status=1
}
build tsan "-fsanitize=thread"
build asan "-fsanitize=address,undefined -fno-sanitize-recover=undefined"
T="$OUT/r06_tsan.txt"; A="$OUT/r06_asan.txt"
{ echo "# tools/run_sanitizers.sh: ThreadSanitizer, $(g++ --version | head -1), $(nproc) CPUs, $SECONDS_EACH s per run"; } > "$T"
{ echo "# tools/run_sanitizers.sh: AddressSanitizer + UBSan, $(g++ --version | head -1), $(nproc) CPUs, $SECONDS_EACH s per run"; } > "$A"
export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4"
export ASAN_OPTIONS="detect_leaks=1 detect_stack_use_after_return=1 strict_string_checks=1"
export UBSAN_OPTIONS="print_stacktrace=1"
# Kernel arguments in host memory, as bench.py runs: the movers' argument blocks
# then go through the per-stream argument rings (device_rings.h ArgRing).
export HIP_FORCE_DEV_KERNARG=0
for sel in uniform prioritized; do
  run "$T" "$WORK/soak_tsan" --seconds "$SECONDS_EACH" --selector $sel --samplers 4
  run "$A" "$WORK/soak_asan" --seconds "$SECONDS_EACH" --selector $sel --samplers 4 --fork
done
  # two paths, one history: plain adds against early insert + helper thread +
  # predicted rows + carried writes, every sampled batch equal byte for byte
for sel in uniform prioritized; do
  run "$T" "$WORK/soak_tsan" --compare 1500 --selector $sel
  run "$A" "$WORK/soak_asan" --compare 3000 --selector $sel
done
EMB_DEFER_INDEX=0 run "$T" "$WORK/soak_tsan" --seconds 5 --selector uniform --samplers 2 --no-deferred-check
EMB_PREDICT_ROWS=0 run "$T" "$WORK/soak_tsan" --seconds 5 --selector uniform --samplers 2
HIP_FORCE_DEV_KERNARG=1 run "$T" "$WORK/soak_tsan" --seconds 5 --selector uniform --samplers 4
HIP_FORCE_DEV_KERNARG=1 run "$A" "$WORK/soak_asan" --seconds 5 --selector prioritized --samplers 4
for f in "$T" "$A"; do
  n=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:\|ERROR: LeakSanitizer" "$f")
  echo "reports: $n" >> "$f"
  [ "$n" -ne 0 ] && status=1
done
rm -rf "$WORK"
tail -n 3 "$T"; tail -n 3 "$A"
exit $status
