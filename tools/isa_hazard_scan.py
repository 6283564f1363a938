"""Scan gfx950 ISA text (hipcc -save-temps: *-gfx950.s) for the two documented "VALU writes an SGPR" hazards the compiler must pad:
VALU-written SGPR -> v_readlane / v_writelane lane select (4 wait states) and VALU-written SGPR -> VMEM reads it (5 wait states).
Prints every violation inside a basic block (none on the shipped heads kernel).  usage: isa_hazard_scan.py file.s"""
import re,sys
def scan(fn):
    lines=[l.rstrip() for l in open(fn)]
    ins=[(i,l.strip()) for i,l in enumerate(lines) if l.startswith('\t') and not l.strip().startswith('.') ]
    out=[]
    def sregs(tok):
        m=re.match(r's\[(\d+):(\d+)\]',tok)
        if m: return set(range(int(m.group(1)),int(m.group(2))+1))
        m=re.match(r's(\d+)$',tok)
        if m: return {int(m.group(1))}
        if tok=='vcc': return {'vcc'}
        return set()
    for k,(i,l) in enumerate(ins):
        op=l.split()[0]
        if not op.startswith('v_'): continue
        args=[a.strip() for a in l[len(op):].split(',')]
        written=set()
        if op in('v_readlane_b32','v_readfirstlane_b32'): written=sregs(args[0])
        elif op.startswith('v_cmp') and not op.startswith('v_cmpx'):
            written=sregs(args[0]) if (args[0].startswith('s') or args[0]=='vcc') else {'vcc'}
        elif op.startswith('v_mad_i64') or op.startswith('v_mad_u64') or op.startswith('v_add_co') or op.startswith('v_sub_co') or op.startswith('v_addc') or op.startswith('v_div_scale'):
            written=sregs(args[1])
        if not written: continue
        ws=0
        for (i2,l2) in ins[k+1:k+12]:
            op2=l2.split()[0]
            a2=[a.strip() for a in l2[len(op2):].split(',')]
            if op2.startswith('s_cbranch') or op2=='s_branch' or op2.startswith('s_setpc'): break
            used=set()
            if op2 in('v_readlane_b32','v_writelane_b32') and len(a2)>=3: 
                used=sregs(a2[2]); need=4
            elif op2.startswith(('global_','buffer_','flat_','scratch_')):
                used=set().union(*[sregs(t) for t in a2]); need=5
            else: need=0
            if used & written and ws<need:
                out.append((i+1,l,i2+1,l2,ws,need))
            # count wait states
            if op2=='s_nop': ws+=int(a2[0])+1
            else: ws+=1
            # overwritten by another?
    return out
for r in scan(sys.argv[1]): print(r)
