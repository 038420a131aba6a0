import csv, collections, sys, os
d = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else 'conv_f32'   # substring of the kernel name to aggregate
agg = collections.defaultdict(list)
for p in ('p1', 'p2'):
    f = os.path.join(d, p, p + '_counter_collection.csv')
    for r in csv.DictReader(open(f)):
        if KERNEL in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v[1:]) / len(v[1:]) for k, v in agg.items()}
kt = list(csv.DictReader(open(os.path.join(d, 'p1', 'p1_kernel_trace.csv'))))
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in kt if KERNEL in r['Kernel_Name']][1:]
us = sum(dur) / len(dur)
cyc = m['GRBM_GUI_ACTIVE'] / 8
print(open(os.path.join(d, 'plain.log')).read().strip())
print("profiled duration %.1f us, clock %.2f GHz" % (us, cyc / us / 1e3))
print("MFMA util %.1f%%  (busy %.3g / (1024 SIMD x %.3g cyc))" % (100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc, m['SQ_VALU_MFMA_BUSY_CYCLES'], cyc))
wc = m['SQ_WAVE_CYCLES']
print("wave-cycles: WAIT_ANY %.1f%%  WAIT_INST_ANY %.1f%%  ACTIVE_INST_ANY %.1f%%" % (100 * m['SQ_WAIT_ANY'] / wc, 100 * m['SQ_WAIT_INST_ANY'] / wc, 100 * m['SQ_ACTIVE_INST_ANY'] / wc))
for k in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS'):
    if k in m: print("  %-24s %.4g" % (k, m[k]))
