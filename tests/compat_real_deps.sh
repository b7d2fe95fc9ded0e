#!/usr/bin/env bash
# Builds ro-map_amd/compat/ (the source-compatible nerf:: classes) + tests/compat_driver.cpp against the REAL Eigen3 / OpenCV / GLEW of the machine -- the types
# the reference's consumers pass (CORE/include/common.h:25-30 Eigen::Vector3f members, nerf_manager.h:9-10 <opencv/cv.hpp>, Eigen::Matrix4f by const-ref) -- where
# the image only has tests/compat_stubs/.  The static_asserts of mon_compat.cpp (POD sizes / offsets, column-major 16-float Matrix4f) are then checked against
# the real headers, and the driver runs the consumers' call sequences if a sequence is given.
#   tests/compat_real_deps.sh [<sequence dir> <network json> <out dir>]      exit 0 built (and ran), 77 skipped: a dependency is missing, else failure
set -u
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
find_inc() { for d in "$@"; do [ -n "$d" ] && [ -e "$d" ] && { echo "$d"; return 0; }; done; return 1; }
EIGEN_INC="${EIGEN3_INCLUDE_DIR:-}"
[ -z "$EIGEN_INC" ] && for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/homebrew/include/eigen3 /opt/conda/include/eigen3; do [ -e "$d/Eigen/Core" ] && EIGEN_INC="$d" && break; done
if [ -z "$EIGEN_INC" ] && command -v cmake > /dev/null; then
  f=$(cd /tmp && cmake --find-package -DNAME=Eigen3 -DCOMPILER_ID=GNU -DLANGUAGE=CXX -DMODE=COMPILE 2>/dev/null | tr ' ' '\n' | sed -n 's/^-I//p' | head -1); [ -n "$f" ] && [ -e "$f/Eigen/Core" ] && EIGEN_INC="$f"
fi
CV_CFLAGS=""; CV_LIBS=""
if command -v pkg-config > /dev/null; then
  for pc in opencv opencv4; do pkg-config --exists $pc 2>/dev/null && { CV_CFLAGS="$(pkg-config --cflags $pc)"; CV_LIBS="$(pkg-config --libs-only-L $pc) -lopencv_core"; break; }; done
fi
if [ -z "$CV_CFLAGS" ]; then
  for d in /usr/include /usr/local/include /usr/include/opencv4 /usr/local/include/opencv4; do [ -e "$d/opencv2/core.hpp" ] && { CV_CFLAGS="-I$d"; CV_LIBS="-lopencv_core"; break; }; done
fi
GLEW_INC=""; for d in /usr/include /usr/local/include; do [ -e "$d/GL/glew.h" ] && GLEW_INC="$d" && break; done
missing=""
[ -z "$EIGEN_INC" ] && missing="$missing Eigen3"; [ -z "$CV_CFLAGS" ] && missing="$missing OpenCV"; [ -z "$GLEW_INC" ] && missing="$missing GLEW"
if [ -n "$missing" ]; then echo "compat_real_deps: skipped, not on this machine:$missing"; exit 77; fi
[ -e "$HERE/ro-map_amd/libmon_core.so" ] || { echo "compat_real_deps: build libmon_core.so first (python -c 'import __graft_entry__ as g; g.build()')"; exit 1; }
OUT="${TMPDIR:-/tmp}/compat_real_deps.$$"; mkdir -p "$OUT"
set -x
g++ -std=c++14 -O1 -Wall -Wextra -I"$EIGEN_INC" $CV_CFLAGS -I"$GLEW_INC" -I"$HERE/ro-map_amd/compat" -I"$HERE/include" \
    "$HERE/ro-map_amd/compat/mon_compat.cpp" "$HERE/tests/compat_driver.cpp" -o "$OUT/compat_driver" \
    -L"$HERE/ro-map_amd" -lmon_core -Wl,-rpath,"$HERE/ro-map_amd" $CV_LIBS -lpthread || { set +x; echo "compat_real_deps: BUILD FAILED against the real headers"; exit 1; }
set +x
echo "compat_real_deps: built against Eigen ($EIGEN_INC), OpenCV ($CV_CFLAGS), GLEW ($GLEW_INC)"
if [ $# -ge 3 ]; then
  "$OUT/compat_driver" offline "$1" "$2" "$3" || { echo "compat_real_deps: offline sequence FAILED"; exit 1; }
  "$OUT/compat_driver" online "$1" "$2" "$3.online" || { echo "compat_real_deps: online sequence FAILED"; exit 1; }
fi
rm -rf "$OUT"; exit 0
