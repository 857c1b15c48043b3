import json, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
print('value', round(a['value'], 1), round(b['value'], 1), 'ms', round(a['ms_per_step'], 3), round(b['ms_per_step'], 3), 'serial', round(a['kernels']['serial_ms_per_step'], 3), round(b['kernels']['serial_ms_per_step'], 3))
ka = {(k['entry'], str(k['args'])): k for k in a['kernels']['top']}
kb = {(k['entry'], str(k['args'])): k for k in b['kernels']['top']}
for key in ka:
    if key in kb and abs(ka[key]['ms_per_step'] - kb[key]['ms_per_step']) > 0.01:
        print('%-40s %-34s %.3f -> %.3f' % (key[0], key[1][:34], ka[key]['ms_per_step'], kb[key]['ms_per_step']))
for key in kb:
    if key not in ka: print('only in b', key, round(kb[key]['ms_per_step'], 3))
for key in ka:
    if key not in kb: print('only in a', key, round(ka[key]['ms_per_step'], 3))
