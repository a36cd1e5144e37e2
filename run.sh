#!/usr/bin/env bash
# Same surface as the reference's run.sh (hw8/run.sh:2-9):
#   ./run.sh <scene.gltf> <width> <height> <samples> <out.ppm> [<envmap.png>]
if [ $# -eq 6 ]
then
    echo "Launching version with environment map"
    ./build/main "$1" "$2" "$3" "$4" "$5" "$6"
else
    echo "Launching version without environment map"
    ./build/main "$1" "$2" "$3" "$4" "$5"
fi;
