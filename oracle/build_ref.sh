#!/bin/sh
# Compile the reference's OWN hot-path functions into oracle/_ref/libsdrref.so.
#
# The reference cannot be built as a whole here (it needs libbladeRF, OpenCV, ArrayFire and the
# un-vendored davemers0160/Common headers), but the DSP functions in
# bladerf/apt_demod/src/bladerf_apt_demod.cpp and bladerf/common/include/sdr_functions.h are
# self-contained std::vector code.  This script pulls those function bodies, by line range,
# from the sources where they lie under $REF into a throw-away translation unit in a temp
# directory (never into the repository), compiles it together with ref_shim.cpp (our extern "C"
# wrapper, which contains no reference code) and leaves only the shared object in oracle/_ref/.
# Nothing is stubbed: the prelude consists of standard headers plus the `using namespace std`
# that the reference inherits from its Common includes.
set -eu
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
APT="$REF/bladerf/apt_demod/src/bladerf_apt_demod.cpp"
FUN="$REF/bladerf/common/include/sdr_functions.h"
if [ ! -f "$APT" ] || [ ! -f "$FUN" ]; then
    echo "build_ref.sh: reference not present at $REF (expected on the GPU box); skipping" >&2
    exit 0
fi
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
# line ranges: polyphase_decimate 249-313; vec_ptws_mul/decimate_vec/
# polar_discriminator/filter_vec 351-455; frequency_shift 476-495; create_freq_rotation 63-74.
{
    sed -n '249,313p' "$APT"
    sed -n '351,455p' "$APT"
    sed -n '476,495p' "$APT"
    sed -n '63,74p' "$FUN"
} > "$TMP/ref_extract.inc"
# sanity: the extraction must contain the functions we expect (guards against a shifted tree)
for sym in polyphase_decimate polar_discriminator decimate_vec filter_vec frequency_shift create_freq_rotation; do
    grep -q "$sym" "$TMP/ref_extract.inc" || { echo "build_ref.sh: $sym not found in extraction" >&2; exit 1; }
done
mkdir -p "$HERE/_ref"
g++ -O2 -std=c++17 -fPIC -shared -I"$TMP" -o "$HERE/_ref/libsdrref.so" "$HERE/ref_shim.cpp"
echo "built $HERE/_ref/libsdrref.so"
