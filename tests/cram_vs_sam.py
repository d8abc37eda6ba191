#!/usr/bin/env python3
"""tests/cram_vs_sam.py — TEST INFRASTRUCTURE (dev container: needs oracle/_ref/sam2bam, i.e. the reference's htslib).
Differential test of the product's CRAM 3.0 reader (pandepth_amd/host/cram.cpp) against the SAM text a CRAM was written
from: random alignments (CIGARs with M I D N S H = X, bases with mismatches, qualities, tags, mate fields, unmapped
reads, tiny and large contigs -> single- and multi-reference slices, up to 120 000 records -> several containers, raw /
gzip / rANS order-0 / order-1 blocks; 1 to 10 000 records per slice, several slices per container, forced
multi-reference slices, embedded reference) are written as CRAM by htslib — reference-free or against a FASTA — and
tests/harness/cram_check must print the same (tid, pos, flag, mapq, reference-consuming CIGAR shape) for both files.
(mapq of reads flagged unmapped is not compared: CRAM does not store it.)

    python tests/cram_vs_sam.py <seed> <cases>
"""
import random, subprocess, sys, os, re
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S2B=os.path.join(ROOT,'oracle','_ref','sam2bam'); CHK=os.path.join(ROOT,'tests','harness','cram_check')
def gen(rng, n, with_seq, sorted_):
    contigs=[('c%d'%k, rng.choice([1,300,2000,50000])) for k in range(rng.randrange(1,6))]
    if all(l<300 for _,l in contigs): contigs.append(('big', 5000))
    ref={nm:''.join(rng.choice('ACGT') for _ in range(l)) for nm,l in contigs}
    recs=[]
    for i in range(n):
        ci=rng.randrange(len(contigs)); nm,L=contigs[ci]
        if L<300: continue
        ops=[]; k=rng.randrange(1,6)
        for j in range(k):
            ops.append((rng.randrange(1,60), 'M'))
            if j<k-1: ops.append((rng.randrange(1,30), rng.choice('IDNMX=DI')))
        if rng.random()<0.3: ops.insert(0,(rng.randrange(1,20),'S'))
        if rng.random()<0.3: ops.append((rng.randrange(1,20),'S'))
        if rng.random()<0.1: ops.insert(0,(rng.randrange(1,20),'H'))
        # merge adjacent same ops
        m=[]
        for ln,op in ops:
            if m and m[-1][1]==op: m[-1]=(m[-1][0]+ln,op)
            else: m.append((ln,op))
        span=sum(l for l,o in m if o in 'MDN=X')
        if span>=L: continue
        pos=rng.randrange(1,L-span+1)
        rl=sum(l for l,o in m if o in 'MIS=X')
        flag=rng.choice([0,16,99,147,83,163,256,1024,2048,0,16])
        cig=''.join('%d%s'%x for x in m)
        if with_seq:
            # read bases: reference bases with some mismatches
            seq=[]; rp=pos-1
            for l,o in m:
                if o in 'M=X':
                    for t in range(l):
                        b=ref[nm][rp+t]
                        if o=='X' or (o=='M' and rng.random()<0.05): b=rng.choice([x for x in 'ACGT' if x!=b])
                        seq.append(b)
                    rp+=l
                elif o in 'IS': seq += [rng.choice('ACGTN') for _ in range(l)]
                elif o in 'DN': rp+=l
            seq=''.join(seq); qual=''.join(chr(33+rng.randrange(2,41)) for _ in seq)
        else: seq='*'; qual='*'
        tags=rng.choice(['', '\tNM:i:%d'%rng.randrange(9), '\tRG:Z:g1\tAS:i:%d'%rng.randrange(200), '\tXA:Z:foo,bar;\tMD:Z:10A5'])
        mate = ('=',pos+rng.randrange(0,300),rng.randrange(-400,400)) if flag&1 else ('*',0,0)
        recs.append((ci,pos,'r%d\t%d\t%s\t%d\t%d\t%s\t%s\t%d\t%d\t%s\t%s%s'%(i,flag,nm,pos,rng.choice([0,3,20,60]),cig,mate[0],mate[1],mate[2],seq,qual,tags)))
    for i in range(rng.randrange(0,4)):
        recs.append((10**6,0,'u%d\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s'%(i,rng.choice([4,77,141]),'ACGTN'*4 if with_seq else '*','IIIII'*4 if with_seq else '*')))
    if sorted_: recs.sort(key=lambda r:(r[0],r[1]))
    hdr='@HD\tVN:1.6\tSO:%s\n'%('coordinate' if sorted_ else 'unsorted')+''.join('@SQ\tSN:%s\tLN:%d\n'%c for c in contigs)+'@RG\tID:g1\tSM:x\n'
    return hdr+'\n'.join(r[2] for r in recs)+'\n', ref
def norm(txt):
    out=[]
    for l in txt.split('\n'):
        f=l.split('\t')
        if len(f)==5 and int(f[2])&4: f[3]='-'
        out.append('\t'.join(f))
    return out
seed=int(sys.argv[1]); cases=int(sys.argv[2]); bad=0
rng=random.Random(seed)
import tempfile
WORK=tempfile.mkdtemp(prefix='cramvs',dir='/tmp'); os.chdir(WORK)
for k in range(cases):
    with_seq=rng.random()<0.7; sorted_=rng.random()<0.8
    n=rng.choice([20,300,300,3000,30000]) if k%10 else 120000
    sam,ref=gen(rng,n,with_seq,sorted_)
    open('t.sam','w').write(sam)
    args=[S2B,'t.sam','t.cram','noindex']
    useref = with_seq and sorted_ and rng.random()<0.6
    if useref:
        open('t.fa','w').write(''.join('>%s\n%s\n'%(nm,sq) for nm,sq in ref.items()))
        if os.path.exists('t.fa.fai'): os.remove('t.fa.fai')
        args.append('ref=t.fa')
    if rng.random()<0.4: args.append('sps=%d'%rng.choice([1,7,100,1000]))
    if rng.random()<0.4: args.append('spc=%d'%rng.choice([2,3,10]))
    if rng.random()<0.2: args.append('multiseq')
    if useref and rng.random()<0.3: args.append('embedref')
    u=rng.random()
    if u<0.45: args.append('fmt=cram,version=3.1'+rng.choice(['',',fast',',normal',',level=1',',level=9']))    # rANS Nx16 blocks
    elif u<0.6: args.append('fmt=cram,version=2.1')                                                       # the older framing
    p=subprocess.run(args,capture_output=True)
    if p.returncode: print('case',k,'s2b failed',p.stderr.decode()[-200:]); continue
    a=subprocess.run([CHK,'t.cram',str(rng.choice([1,1,3,8]))],capture_output=True); b=subprocess.run([CHK,'t.sam'],capture_output=True)
    if a.returncode or norm(a.stdout.decode())!=norm(b.stdout.decode()):
        bad+=1; print('MISMATCH case',k,'n',n,'seq',with_seq,'sorted',sorted_,'ref',useref,' '.join(args[4:]),a.stderr.decode()[-200:]); 
        os.system('cp t.sam /tmp/crambad_%d_%d.sam; cp t.cram /tmp/crambad_%d_%d.cram'%(seed,k,seed,k))
import shutil
shutil.rmtree(WORK,ignore_errors=True)
print('seed %d: %d cases, %d mismatches'%(seed,cases,bad))
sys.exit(1 if bad else 0)
