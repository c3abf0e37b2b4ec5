#!/bin/bash
# Sanitizer pass over the HOST side on the CPU build (no GPU sanitizer, no XNACK: not available on this pool).
#   bash scripts/sanitize_cpu.sh [asan|tsan|tsan-lanes|all]        logs -> profiles/<round>_sanitize_<mode>.log
# Works on a scratch COPY of the tree (default /tmp/kzg_san/<mode>): the shipped .so files are never touched.
# What is instrumented, with ONE runtime (the ROCm clang's, preloaded into the uninstrumented python):
#   - libkzg_mi355x.so: every host translation unit (the host side of csrc/lanes / srs / pipeline / serve / comm .hip, multi_host.cpp, pairing_host.cpp, finish_host.cpp, wire_host.cpp) via
#     hipcc -fsanitize=... -fno-gpu-sanitize (device code stays as shipped)
#   - zkp_subnet_amd/_wire (csrc/wire_py.c) and oracle/libkzg_oracle.so (oracle/kzg_cpu.c) via clang
# What runs: asan (ASan + UBSan): the whole CPU suite (pytest -m "not gpu") + tests/san_drive.py;
#            tsan: tests/san_drive.py (8 Python threads on the wire pool incl. the asynchronous batches, the verifier's
#            thread pool, the oracle's task pool) + the threaded CPU tests (test_verify, test_host_logic, test_oracle);
#            tsan-lanes: tests/lanebook_tsan.cpp -- the context's lane / ticket / staging / row-cache state machine
#            (csrc/lanebook.h, the HIP-free half of csrc/lanes.hip) driven by 12 threads with a fake back end and injected
#            failures.  On this box kzg_create answers KZG_E_HIP, so the instrumented libkzg_mi355x.so never reaches that
#            code: the drive is how a race detector gets to see it.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
MODE=${1:-all}
WORK=${SAN_WORK:-/tmp/kzg_san}
CLANG=/opt/rocm/lib/llvm/bin/clang
ROUND=${ROUND:-r06}
run_mode() {
  local mode=$1 san rt opts
  if [ "$mode" = asan ]; then
    san="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
    rt=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
  else
    san="-fsanitize=thread"
    rt=$($CLANG -print-file-name=libclang_rt.tsan-x86_64.so)
  fi
  local W=$WORK/$mode LOG=$R/profiles/${ROUND}_sanitize_$mode.log
  rm -rf "$W"; mkdir -p "$W"
  (cd "$R" && tar cf - --exclude=.git --exclude=gpurun_out --exclude='*.so' --exclude='*.o' --exclude=build \
      --exclude=build_proto --exclude=__pycache__ --exclude=.pytest_cache --exclude=.hypothesis .) | (cd "$W" && tar xf -)
  cd "$W"
  export KZG_EXTRA_HIPCC_FLAGS="$san -fno-gpu-sanitize -shared-libsan -g -fno-omit-frame-pointer"
  export KZG_WIRE_CC=$CLANG KZG_WIRE_CFLAGS="-O1 -g -fno-omit-frame-pointer $san -shared-libsan"
  {
    echo "== $mode build: $(date -u +%FT%TZ)  clang: $($CLANG --version | head -1)"
    echo "== flags: $san   runtime: $rt"
    make -s -C oracle CC=$CLANG CFLAGS="-O1 -g -fno-omit-frame-pointer -march=x86-64-v3 -fPIC $san -shared-libsan" libkzg_oracle.so || exit 1
    python -m zkp_subnet_amd.build --force || exit 1
    for so in zkp_subnet_amd/libkzg_mi355x.so zkp_subnet_amd/_wire*.so oracle/libkzg_oracle.so; do
      echo "instrumented: $so -> needs $(ldd $so | grep -o 'libclang_rt[^ ]*' | head -1), $(nm -D $so | grep -c '__asan_\|__tsan_\|__ubsan_') sanitizer symbol references"
    done
  } > "$LOG" 2>&1
  export LD_PRELOAD=$rt
  export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:detect_stack_use_after_return=1
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  export TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1:report_signal_unsafe=0:exitcode=66"
  local rc=0
  {
    # the harness must SEE a bug of its kind before a clean log means anything
    $CLANG -O1 -g -fPIC -shared -pthread $san -shared-libsan "$R/scripts/san_canary.c" -o "$W/san_canary.so"
    local fn=canary_race; [ "$mode" = asan ] && fn="canary_overflow(1)" || fn="canary_race()"
    if python -c "import ctypes; ctypes.CDLL('$W/san_canary.so').$fn" 2>&1 | grep -q "heap-buffer-overflow\|ThreadSanitizer: data race"; then
      echo "== canary: the deliberate bug of scripts/san_canary.c IS reported by this harness"
    else
      echo "== canary: NOT reported -- the harness is blind, a clean log below proves nothing"; rc=1
    fi
    echo "== tests/san_drive.py (8 Python threads)"
    python tests/san_drive.py ${SAN_ITER:-4} || rc=1
    if [ "$mode" = asan ]; then
      echo "== pytest -m 'not gpu' (the whole CPU suite)"
      python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider || rc=1
    else
      echo "== pytest: the threaded CPU tests"
      python -m pytest tests/test_verify.py tests/test_host_logic.py tests/test_oracle.py -x -q -p no:cacheprovider || rc=1
    fi
    echo "== reports outside the canary: $(grep -c 'ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer' "$LOG") (0 = clean)  rc=$rc"
  } >> "$LOG" 2>&1
  unset LD_PRELOAD
  cd "$R"
  tail -n 4 "$LOG"
  return $rc
}
run_lanes() {
  local W=$WORK/tsan-lanes LOG=$R/profiles/${ROUND}_sanitize_tsan_lanes.log rc=0
  rm -rf "$W"; mkdir -p "$W"
  {
    echo "== tsan-lanes build: $(date -u +%FT%TZ)  clang: $($CLANG --version | head -1)"
    ${CLANG}++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=thread -pthread -I "$R/zkp_subnet_amd/csrc" \
        "$R/tests/lanebook_tsan.cpp" -o "$W/lanebook_tsan" || exit 1
    export TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1:exitcode=66"
    if "$W/lanebook_tsan" canary 2>&1 | grep -q "ThreadSanitizer: data race"; then
      echo "== canary: two threads touching one lane without the book ARE reported by this harness"
    else
      echo "== canary: NOT reported -- the harness is blind, a clean log below proves nothing"; rc=1
    fi
    echo "== tests/lanebook_tsan.cpp: ${LANES_S:-20} s, ${LANES_THREADS:-12} threads"
    "$W/lanebook_tsan" ${LANES_S:-20} ${LANES_THREADS:-12} > "$W/drive.out" 2>&1 || rc=1
    cat "$W/drive.out"
    echo "== ThreadSanitizer reports in the drive: $(grep -c 'WARNING: ThreadSanitizer' "$W/drive.out") (0 = clean)  rc=$rc"
  } > "$LOG" 2>&1
  tail -n 3 "$LOG"
  return $rc
}
status=0
case $MODE in
  asan|tsan) run_mode $MODE || status=1 ;;
  tsan-lanes) run_lanes || status=1 ;;
  all) run_mode asan || status=1; run_mode tsan || status=1; run_lanes || status=1 ;;
  *) echo "usage: $0 [asan|tsan|tsan-lanes|all]"; exit 2 ;;
esac
exit $status
