# round 4: kernel trace of two C4 bf16 steps with the side stream off: per-dispatch durations of the 64x64-tile launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/kt
SDM_WGRAD_STREAM=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/train_probe.py c4 bf16 2 > gpurun_out/kt.log 2>&1 || { tail -5 gpurun_out/kt.log; exit 1; }
f=$(ls -t gpurun_out/kt/*/*kernel_trace.csv | head -1)
python - "$f" > gpurun_out/r04_c4_bf16_trace_window.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# windows of 12 dispatches in front of every 64x64-tile launch longer than 120 us (last step only)
n = len(rows)
t0 = int(rows[0]["Start_Timestamp"])
for i in range(n * 2 // 3, n):
    r = rows[i]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "igemm_bf16_kernel<64" in r["Kernel_Name"] and dur > 120:
        print("---- slow launch at dispatch", i)
        for j in range(i - 12, i + 3):
            q = rows[j]
            d = (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3
            gap = (int(q["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3
            gx = int(q["Grid_Size_X"]) // max(1, int(q["Workgroup_Size_X"]))
            print(f"{j:6d} gap {gap:8.1f} dur {d:8.1f} grid {gx:6d} {q['Grid_Size_Y']:>3s} {q['Grid_Size_Z']:>3s} lds {q.get('LDS_Block_Size','?'):>7s} {q['Kernel_Name'][:90]}")
PY
rm -rf gpurun_out/kt
SDM_WGRAD_STREAM=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/train_probe.py c4 bf16 2 > gpurun_out/kt.log 2>&1 || { tail -5 gpurun_out/kt.log; exit 1; }
f=$(ls -t gpurun_out/kt/*/*kernel_trace.csv | head -1)
python - "$f" > gpurun_out/r04_c4_bf16_trace_small_tiles.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("# kernel, grid (x,y,z workgroups), duration us -- last C4 bf16 step, side stream off; 64x64 tiles, deep-level launches")
n = len(rows)
last = rows[n // 2 + n // 4:]          # roughly the last step
for r in rows[-700:]:
    name = r["Kernel_Name"]
    if "igemm_bf16_kernel<64" in name or "splitk_epilogue_bf16" in name or "igemm_bf16_kernel<128, 128, 2, 2, 2, true" in name:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gx = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])); gy = int(r["Grid_Size_Y"]); gz = int(r["Grid_Size_Z"])
        print(f"{name[:60]:60s} {gx:6d} {gy:3d} {gz:3d} {dur:9.1f}")
PY
rm -rf gpurun_out/kt
cat gpurun_out/r04_c4_bf16_trace_window.txt | cut -c1-200
