#!/usr/bin/env bash
# Same surfaces as the reference's run.sh scripts:
#   hw8/run.sh:2-9   ./run.sh <scene.gltf> <width> <height> <samples> <out.ppm> [<envmap.png>]
#   hw1/run.sh:2     ./run.sh <scene.txt> <out.ppm>
# RTAMD_SNAPSHOT=hw1|hw3|hw6|hw8 picks which snapshot's integrator replays the scene (default hw8 / hw3).
if [ $# -eq 2 ]
then
    ./build/main "$1" "$2"
elif [ $# -eq 6 ]
then
    echo "Launching version with environment map"
    ./build/main "$1" "$2" "$3" "$4" "$5" "$6"
else
    echo "Launching version without environment map"
    ./build/main "$1" "$2" "$3" "$4" "$5"
fi;
