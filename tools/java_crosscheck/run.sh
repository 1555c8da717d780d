#!/bin/bash
# One command: pin the CPU oracle to the reference's own classes.  Needs `java`, `javac` and the reference + its four jars on
# MMIDX_REFERENCE_CLASSPATH (see README.md).  Exit 0 = every fixture's ids and distance bits agree (prints "CROSSCHECK OK"),
# 2 = no JDK / classpath (nothing was run), 1 = a mismatch or a failure.
#   MMIDX_REFERENCE_CLASSPATH=ref.jar:lingpipe-4.0.1.jar:je-5.0.58.jar:ejml-0.23.jar:trove4j-3.0.3.jar tools/java_crosscheck/run.sh [workdir]
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
WORK=${1:-/tmp/mmidx_crosscheck}
CP=${MMIDX_REFERENCE_CLASSPATH:-}
command -v java >/dev/null 2>&1 && command -v javac >/dev/null 2>&1 || { echo "CROSSCHECK SKIPPED: no JDK on this box"; exit 2; }
[ -n "$CP" ] || { echo "CROSSCHECK SKIPPED: MMIDX_REFERENCE_CLASSPATH is not set (reference jar + lingpipe / je / ejml / trove4j)"; exit 2; }
for j in ${CP//:/ }; do [ -e "$j" ] || { echo "CROSSCHECK SKIPPED: $j does not exist"; exit 2; }; done
mkdir -p "$WORK/fixtures" "$WORK/classes" "$WORK/out" || exit 1
python "$HERE/export_fixtures.py" "$WORK/fixtures" || exit 1
javac -cp "$CP" -d "$WORK/classes" "$HERE/CrossCheck.java" || exit 1
java -cp "$CP:$WORK/classes" CrossCheck "$WORK/fixtures" "$WORK/out" || exit 1
python "$HERE/compare.py" "$WORK/fixtures" "$WORK/out" && echo "CROSSCHECK OK"
