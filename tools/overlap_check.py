import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')
rows = list(csv.DictReader(open(f[0])))
convs = [r for r in rows if 'conv_f32' in r['Kernel_Name']]
convs.sort(key=lambda r: int(r['Start_Timestamp']))
print(len(convs), 'queues', set(r['Queue_Id'] for r in convs), 'streams', set(r['Stream_Id'] for r in convs))
mid = convs[len(convs) // 2: len(convs) // 2 + 400]
ov = sum(1 for a, b in zip(mid[:-1], mid[1:]) if int(b['Start_Timestamp']) < int(a['End_Timestamp']))
print('overlapping consecutive pairs', ov, 'of', len(mid) - 1)
t0 = int(mid[0]['Start_Timestamp'])
for r in mid[:10]:
    print(r['Queue_Id'], r['Stream_Id'], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r['Grid_Size_X'])
