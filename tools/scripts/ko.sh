export MS3D_WIDE=0
for kb in 0 78 52; do for w in 8 5 4; do
  MS3D_STREAM_LDS_KB=$kb MS3D_STREAM_WAVES=$w python tools/conv_micro.py 64 64 27 1 2>&1 | grep cin | cut -c60-200
done; done
for kb in 78 52; do for w in 8 5; do
  MS3D_STREAM_LDS_KB=$kb MS3D_STREAM_WAVES=$w python tools/conv_micro.py 96 96 27 2 2>&1 | grep cin | cut -c60-200
done; done
