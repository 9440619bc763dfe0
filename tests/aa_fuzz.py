"""tests/aa_fuzz.py — random AirAssembly modules for comparing the two loaders (js/aa_loader.js, genstark_amd/airassembly.py):
1 - 4 registers, scalar / vector / matrix constants, literal / power / prng cycles, a function, locals, and expression trees over
add sub mul div exp prod vector get slice call load.{trace, static, const, param, local} with scalar-vector broadcasting."""
P = 340282366920938463463374607393113505793


def gen_module(rnd):
    R = rnd.choice([1,2,3,4])
    ncyc = rnd.choice([0,1,2])
    nconst = rnd.choice([0,1,2,3])
    consts=[]; cdefs=[]
    for i in range(nconst):
        kind = rnd.choice(['scalar','vector','matrix'])
        if kind=='scalar': cdefs.append(f'(const $c{i} scalar {rnd.randrange(1,50)})'); consts.append(('scalar',1))
        elif kind=='vector':
            n=R; cdefs.append(f'(const $c{i} vector {" ".join(str(rnd.randrange(1,99)) for _ in range(n))})'); consts.append(('vector',n))
        else:
            cdefs.append(f'(const $c{i} matrix {" ".join("("+" ".join(str(rnd.randrange(1,9)) for _ in range(R))+")" for _ in range(R))})'); consts.append(('matrix',R))
    cycles=[]
    for i in range(ncyc):
        m = rnd.choice([2,4,8])
        g = rnd.random()
        if g<0.5: cycles.append('(cycle '+' '.join(str(rnd.randrange(1,1000)) for _ in range(m))+')')
        elif g<0.8: cycles.append(f'(cycle (power {rnd.randrange(2,9)} {m}))')
        else: cycles.append(f'(cycle (prng sha256 0x{rnd.randrange(1<<32):08x} {m}))')
    # expression generator: returns (text, width) width None = scalar, else vector length
    def scalar(depth, ctx):
        r = rnd.random()
        if depth<=0 or r<0.25:
            c = rnd.random()
            if c<0.35: return f'(get (load.trace 0) {rnd.randrange(R)})'
            if c<0.5 and ctx.get('next'): return f'(get (load.trace 1) {rnd.randrange(R)})'
            if c<0.65 and ncyc: return f'(get (load.static 0) {rnd.randrange(ncyc)})'
            if c<0.8:
                sc=[i for i,(k,_) in enumerate(consts) if k=='scalar']
                if sc: return f'(load.const $c{rnd.choice(sc)})'
            return f'(scalar {rnd.randrange(0,20)})'
        if r<0.45: return f'({rnd.choice(["add","sub","mul"])} {scalar(depth-1,ctx)} {scalar(depth-1,ctx)})'
        if r<0.55: return f'(exp {scalar(depth-1,ctx)} (scalar {rnd.choice([2,3,3,5])}))'
        if r<0.62: return f'(div {scalar(depth-1,ctx)} (scalar {rnd.randrange(1,9)}))'
        if r<0.75:
            v,w = vector(depth-1,ctx); return f'(get {v} {rnd.randrange(w)})'
        if r<0.85 and ctx.get('fn'):
            return f'(get (call $f {vector(depth-1,ctx,R)[0]} {scalar(depth-1,ctx)}) {rnd.randrange(R)})'
        return f'(mul {scalar(depth-1,ctx)} {scalar(depth-1,ctx)})'
    def vector(depth, ctx, want=None):
        w = want or rnd.choice([1,2,R])
        r = rnd.random()
        if depth<=0 or r<0.3:
            if w==R and rnd.random()<0.5: return '(load.trace 0)', R
            return '(vector '+' '.join(scalar(depth-1,ctx) for _ in range(w))+')', w
        if r<0.5:
            a,_=vector(depth-1,ctx,w); b = scalar(depth-1,ctx) if rnd.random()<0.4 else vector(depth-1,ctx,w)[0]
            return f'({rnd.choice(["add","sub","mul"])} {a} {b})', w
        if r<0.6:
            a,_=vector(depth-1,ctx,w); return f'(exp {a} (scalar {rnd.choice([2,3])}))', w
        if r<0.7 and w==R:
            mats=[i for i,(k,_) in enumerate(consts) if k=='matrix']
            if mats: return f'(prod (load.const $c{rnd.choice(mats)}) {vector(depth-1,ctx,R)[0]})', R
        if r<0.8:
            a,wa=vector(depth-1,ctx,w+rnd.choice([0,1,2]))
            lo=rnd.randrange(wa-w+1); return f'(slice {a} {lo} {lo+w-1})', w
        if r<0.9 and w==R:
            vc=[i for i,(k,n) in enumerate(consts) if k=='vector']
            if vc: return f'(add (load.const $c{rnd.choice(vc)}) {vector(depth-1,ctx,R)[0]})', R
        return '(vector '+' '.join(scalar(depth-1,ctx) for _ in range(w))+')', w
    depth = rnd.choice([2,3,4])
    fn = ''
    fnctx={'fn':False}
    usefn = rnd.random()<0.6
    if usefn:
        # $f(v: vector R, s: scalar) -> vector R
        body = f'(add (mul (load.param $v) (load.param $s)) (exp (load.param $v) (scalar {rnd.choice([2,3])})))'
        fn = f'(function $f (result vector {R}) (param $v vector {R}) (param $s scalar) {body})'
    tctx={'fn':usefn}
    local = rnd.random()<0.5
    tbody = vector(depth,tctx,R)[0]
    if local:
        tr = f'(local $t vector {R}) (store.local $t {vector(depth-1,tctx,R)[0]}) (add (load.local $t) {tbody})'
        trexpr = lambda : tr
    else:
        tr = tbody
    ectx={'fn':usefn,'next':True}
    if local:
        ev = f'(local $t vector {R}) (store.local $t {tr.split("(store.local $t ",1)[1].rsplit(") (add (load.local $t)",1)[0]}) (sub (load.trace 1) (add (load.local $t) {tbody}))'
    else:
        ev = f'(sub (load.trace 1) {tbody})'
    steps = rnd.choice([8,16,32])
    static = f'(static {" ".join(cycles)})' if cycles else ''
    src = f'''(module (field prime {P}) {" ".join(cdefs)} {fn}
  (export main (registers {R}) (constraints {R}) (steps {steps}) {static}
    (init (param $seed vector {R}) (load.param $seed))
    (transition {tr})
    (evaluation {ev})))'''
    return src, R
