#!/bin/bash
# Is a real OpenCV reachable on this box?  (VERDICT r01 "next" item 1a.)  Output is kept under profiles/.
echo "== python cv2"; python -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1
echo "== conda python3.9 cv2"; /opt/conda/bin/python3.9 -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1
echo "== pkg-config"; (pkg-config --modversion opencv4 opencv 2>&1 || true) | tail -2
echo "== headers"; ls -d /usr/include/opencv* /usr/local/include/opencv* /opt/*/include/opencv* 2>&1 | tail -3
echo "== libraries"; (ldconfig -p | grep -i opencv || echo none)
echo "== files named *opencv* / cv2*"; find / -xdev \( -iname "*opencv*" -o -iname "cv2*" -o -iname "*bgfg*" \) -not -path "/proc/*" 2>/dev/null | head -20
echo "== other image libraries"; for m in skimage scipy.ndimage PIL torchvision; do python -c "import $m; print('$m', getattr($m,'__version__',''))" 2>&1 | tail -1; done
/opt/conda/bin/python3.9 -c "import skimage, scipy; print('conda3.9 skimage', skimage.__version__, 'scipy', scipy.__version__)" 2>&1 | tail -1
echo "== host"; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
# ... and if there is one, keep what it says: fixtures for tests/golden/, consumed by tests/test_oracle_golden.py
echo "== harvest"; python "$(dirname "$0")/harvest_opencv_golden.py" --out gpurun_out/opencv_golden 2>&1 | tail -8
