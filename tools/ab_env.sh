#!/bin/bash
# run ab_libs.py with one environment variable set (A/B of a measurement switch): ab_env.sh VAR=VALUE <ab_libs args...>
export "$1"; shift
exec python3 -u "$(dirname "$0")/ab_libs.py" "$@"
