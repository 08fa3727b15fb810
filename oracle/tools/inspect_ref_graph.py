"""Study aid (not shipped logic): parse the reference's Tanner-graph table as TEXT and
report its structural properties, so our own check-major table can be verified against it.
Reads /root/reference (only exists in the build container)."""
import re, sys
import numpy as np

src = open('/root/reference/src/ldpc_context.cuh').read()
body = src[src.index('ldpc_reverse_map[128][3][2]'):src.index('};', src.index('ldpc_reverse_map[128][3][2]'))]
nums = [int(x) for x in re.findall(r'-?\d+', body)][3:]  # drop the 128,3,2 dims
assert len(nums) == 128 * 3 * 2, len(nums)
mp = np.array(nums).reshape(128, 3, 2)  # [bit][edge] = (slot, check)

def report():
    H = np.zeros((38, 128), dtype=np.uint8)
    slot_of = {}
    for n in range(128):
        for k in range(3):
            s, c = mp[n, k]
            assert H[c, n] == 0
            H[c, n] = 1
            slot_of[(c, n)] = s
    print('col weights', set(H.sum(0)), 'row weights', sorted(set(H.sum(1))))
    print('rows with 11:', [c for c in range(38) if H[c].sum() == 11])
    # edge order per bit ascending in check?
    asc = all(mp[n, 0, 1] < mp[n, 1, 1] < mp[n, 2, 1] for n in range(128))
    print('edges per bit ascending by check:', asc)
    if not asc:
        print([ (n, mp[n,:,1].tolist()) for n in range(128) if not (mp[n,0,1] < mp[n,1,1] < mp[n,2,1])])
    # slots per check ascending in bit?
    ok = True
    for c in range(38):
        bits = [n for n in range(128) if H[c, n]]
        slots = [slot_of[(c, n)] for n in bits]
        if slots != list(range(len(bits))):
            ok = False
            print('check', c, 'bits', bits, 'slots', slots)
    print('slots per check ascending by bit:', ok)
    return H

if __name__ == '__main__':
    H = report()
    if len(sys.argv) > 1 and sys.argv[1] == 'emit':
        for c in range(38):
            print('    {' + ', '.join(f'{n:3d}' for n in range(128) if H[c, n]) + '},')
