"""AirAssembly source -> JSON descriptors for the node side (js/shims/@guildofweavers/air-assembly: compile / instantiate).

index.ts:18-33 compiles AirAssembly source with `@guildofweavers/air-assembly` (absent: SURVEY 8c); the loader of this repository is
genstark_amd/airassembly.py.  The node-side shim runs this module as a child process — one JSON request on stdin, one JSON answer on
stdout — instead of carrying a second copy of the loader in JavaScript:

    {"op": "check",    "source": text}                                   -> {"modulus": str, "exports": {name: {"registers", "constraints", "inputs", "secretInputs"}}}
    {"op": "info",     "source", "component", "extensionFactor"}         -> register / constraint counts, degrees, the extension factor in force
    {"op": "describe", "source", "component", "extensionFactor"}         -> {"descriptor": GenericAir.descriptor()}          (components without input registers)
    {"op": "plan",     "source", "component", "extensionFactor", "inputs", "seed"}   -> {"descriptor": ... with firstRows pinned, "inputShapes": [...]}
    {"op": "verify",   "source", "component", "extensionFactor", "inputShapes", "publicInputs"} -> {"descriptor": ...}

Integers travel as decimal strings.  Nothing here touches a device or a library: the field is genstark_amd.hostfield.HostField.
usage: python -m genstark_amd.aa_json < request.json"""
import json
import sys


def _ints(x):
    if isinstance(x, list):
        return [_ints(v) for v in x]
    return int(x)


def handle(req):
    from .airassembly import AssemblyAir, Module, _Layout, _shape_of
    from .hostfield import HostField
    module = Module(req['source'])
    if req['op'] == 'check':
        out = {}
        for name, ex in module.exports.items():
            inputs = [s for s in ex.statics if s['kind'] == 'input']
            out[name] = {'registers': ex.registers, 'constraints': ex.constraints, 'inputs': len(inputs), 'secretInputs': sum(1 for s in inputs if s['secret'])}
        return {'modulus': str(module.modulus), 'exports': out}
    air = AssemblyAir(module, req.get('component') or 'default', req.get('extensionFactor'), field=HostField(module.modulus))
    if req['op'] == 'info':        # what lib/Stark.ts reads of an AirModule before any input arrives (lib/Stark.ts:40-75)
        out = {'traceRegisterCount': air.traceRegisterCount, 'secretInputCount': air.secretInputCount, 'constraintDegrees': air.constraintDegrees,
               'maxConstraintDegree': air.maxConstraintDegree, 'extensionFactor': air.extensionFactor, 'inputRegisters': len(air.inputRegisters)}
        if air.inputRegisters:
            # what the one-call native verify needs of a component with input registers (js/prover.js: verifyAssemblySerialized; struct
            # gs_input_register / gs_static_source of include/gstark_prover.h): the declarations, where each static register of the
            # programs takes its values from, the cyclic registers' values, and the shape-independent constraint evaluator
            sources, cycles = air.staticSources()
            pr = air.evaluationProgram
            out.update({'inputDeclarations': [{'parent': d['parent'], 'peer': d['peer'], 'steps': d['steps'] or 0, 'shift': d['shift'], 'secret': bool(d['secret'])}
                                              for d in air.inputRegisters],
                        'staticSources': [list(x) for x in sources], 'cycles': [[str(v) for v in c] for c in cycles],
                        'evaluation': {'code': [w for ins in pr.code for w in ins], 'consts': [str(v) for v in pr.consts], 'nregs': pr.nregs, 'nout': pr.nout}})
        return out
    if req['op'] == 'describe':
        if air.inputRegisters:
            raise ValueError('the component has input registers: its trace is sized when the inputs arrive')
        inner = air._inner(air._length_without_inputs(), air._public_split(air._columns(_Layout(air.export.statics, []), [])), None)
        return {'descriptor': inner.descriptor()}
    if req['op'] == 'plan':
        seed = req.get('seed')
        inner, packed, firsts, shapes = air.plan(_ints(req.get('inputs') or []), None if seed is None else _ints(seed))
        d = inner.descriptor(seed=firsts)
        d['secretRegisters'] = [[str(v) for v in col.ints()] for col in packed]        # this proof's secret columns (one period each)
        return {'descriptor': d, 'inputShapes': shapes}
    if req['op'] == 'verify':
        shapes = [list(s) for s in (req.get('inputShapes') or [])]
        layout = _Layout(air.export.statics, shapes)
        length = layout.length or air._length_without_inputs()
        public_values, given, j = [], _ints(req.get('publicInputs') or []), 0
        for d in layout.inputs:
            if d['secret']:
                public_values.append(None)
            else:
                public_values.append(given[j])
                j += 1
        inner = air._inner(length, air._public_split(air._columns(layout, public_values)), None)
        return {'descriptor': inner.descriptor()}
    raise ValueError(f'unknown op {req["op"]!r}')


def main():
    try:
        req = json.loads(sys.stdin.read())
        print(json.dumps(handle(req)))
    except Exception as e:   # noqa: BLE001  (the caller turns it into a thrown Error)
        print(json.dumps({'error': f'{type(e).__name__}: {e}'}))


if __name__ == '__main__':
    main()
