"""Mirror of lib/utils/{index,serialization,sizeof,Logger}.ts — host-side helpers and the proof wire format."""
import time

MAX_ARRAY_LENGTH = 256           # sizeof.ts:7
MAX_MATRIX_COLUMN_LENGTH = 127   # sizeof.ts:8


# ---- math (index.ts:13-30)
def isPowerOf2(value):
    return value != 0 and (value & (value - 1)) == 0


def powLog2(base, exponent):
    import math
    twos = 0
    while exponent % 2 == 0:
        twos += 1
        exponent //= 2
    return (2 ** twos) * math.log2(base ** exponent)


# ---- merkle proof (index.ts:34-45)
def rehashMerkleProofValues(proof, hash_):
    return {'nodes': proof['nodes'], 'values': hash_.digestMany(proof['values']), 'depth': proof['depth']}


# ---- big integers (serialization.ts:131-146): LE 32-bit limbs == little-endian bytes
def need(buffer, offset, size):
    """A read of `size` bytes at `offset` must lie inside the buffer (the reference's Buffer reads throw RangeError; a Python
    slice would silently come back short)."""
    if offset < 0 or size < 0 or offset + size > len(buffer):
        from .errors import StarkError
        raise StarkError('malformed proof: truncated')


def readBigInt(buffer, offset, elementSize):
    need(buffer, offset, (elementSize >> 2) * 4)
    return int.from_bytes(buffer[offset:offset + (elementSize >> 2) * 4], 'little')


def writeBigInt(value, buffer, offset, elementSize):
    n = (elementSize >> 2) * 4
    buffer[offset:offset + n] = (value & ((1 << (8 * n)) - 1)).to_bytes(n, 'little')
    return offset + n


# ---- arrays / matrixes (serialization.ts:18-125)
def writeArray(buffer, offset, array):
    buffer[offset] = 0 if len(array) == MAX_ARRAY_LENGTH else len(array)
    offset += 1
    for item in array:
        buffer[offset:offset + len(item)] = item
        offset += len(item)
    return offset


def readArray(buffer, offset, elementSize):
    need(buffer, offset, 1)
    n = buffer[offset] or MAX_ARRAY_LENGTH
    offset += 1
    need(buffer, offset, n * elementSize)
    values = []
    for _ in range(n):
        values.append(bytes(buffer[offset:offset + elementSize]))
        offset += elementSize
    return values, offset


def writeMatrix(buffer, offset, matrix, leafSize):
    buffer[offset] = 0 if len(matrix) == MAX_ARRAY_LENGTH else len(matrix)
    offset += 1
    for column in matrix:
        ctype = 1 if (len(column) > 0 and len(column[0]) == leafSize) else 0  # ColumnType.leaf = 1
        buffer[offset] = ((len(column) << 1) | ctype) & 0xFF
        offset += 1
    for column in matrix:
        for item in column:
            buffer[offset:offset + len(item)] = item
            offset += len(item)
    return offset


def readMatrix(buffer, offset, leafSize, nodeSize):
    need(buffer, offset, 1)
    columnCount = buffer[offset] or MAX_ARRAY_LENGTH
    offset += 1
    need(buffer, offset, columnCount)
    heads = list(buffer[offset:offset + columnCount])
    offset += columnCount
    matrix = []
    for head in heads:
        column = []
        first = leafSize if (head & 1) else nodeSize
        for j in range(head >> 1):
            size = first if j == 0 else nodeSize
            need(buffer, offset, size)
            column.append(bytes(buffer[offset:offset + size]))
            offset += size
        matrix.append(column)
    return matrix, offset


def writeMerkleProof(buffer, offset, proof, leafSize):
    offset = writeArray(buffer, offset, proof['values'])
    offset = writeMatrix(buffer, offset, proof['nodes'], leafSize)
    buffer[offset] = proof['depth']
    return offset + 1


def readMerkleProof(buffer, offset, leafSize, nodeSize):
    values, offset = readArray(buffer, offset, leafSize)
    nodes, offset = readMatrix(buffer, offset, leafSize, nodeSize)
    need(buffer, offset, 1)
    depth = buffer[offset]
    return {'values': values, 'nodes': nodes, 'depth': depth}, offset + 1


# ---- sizes (sizeof.ts:12-99)
def _sizeOfArray(array):
    if len(array) == 0:
        raise ValueError('Array cannot be zero-length')
    if len(array) > MAX_ARRAY_LENGTH:
        raise ValueError(f'Array length ({len(array)}) cannot exceed {MAX_ARRAY_LENGTH}')
    return 1 + sum(len(x) for x in array)


def _sizeOfMatrix(matrix):
    if len(matrix) > MAX_ARRAY_LENGTH:
        raise ValueError(f'Matrix column count ({len(matrix)}) cannot exceed {MAX_ARRAY_LENGTH}')
    size = 1 + len(matrix)
    for column in matrix:
        if len(column) >= MAX_MATRIX_COLUMN_LENGTH:
            raise ValueError(f'Matrix column length ({len(column)}) cannot exceed {MAX_MATRIX_COLUMN_LENGTH}')
        size += sum(len(x) for x in column)
    return size


def sizeOfMerkleProof(proof):
    values, nodes = _sizeOfArray(proof['values']), _sizeOfMatrix(proof['nodes'])
    return {'values': values, 'nodes': nodes, 'total': values + nodes + 1}


def sizeOf(proof, fieldElementSize, hashDigestSize):
    size = hashDigestSize
    evProof = sizeOfMerkleProof(proof['evProof'])
    size += evProof['total']
    ldProof = 1
    lcProof = sizeOfMerkleProof(proof['ldProof']['lcProof'])
    ldProof += lcProof['total'] + hashDigestSize
    levels = []
    for component in proof['ldProof']['components']:
        ldProof += hashDigestSize
        column = sizeOfMerkleProof(component['columnProof'])
        poly = sizeOfMerkleProof(component['polyProof'])
        ldProof += column['total'] + poly['total']
        levels.append({'column': column, 'poly': poly, 'total': column['total'] + poly['total'] + hashDigestSize})
    ldRemainder = len(proof['ldProof']['remainder']) * fieldElementSize + 1
    levels.append({'total': ldRemainder})
    ldProof += ldRemainder
    size += ldProof
    inputShapes = 1 + sum(1 + 4 * len(s) for s in proof['iShapes'])
    size += inputShapes
    return {'evProof': evProof, 'ldProof': {'lcProof': lcProof, 'levels': levels, 'total': ldProof},
            'inputShapes': inputShapes, 'total': size}


# ---- Logger (Logger.ts:8-77): phase timer; the labels are the timing vocabulary of README.md:62-73
class Logger:
    def __init__(self, echo=True, sync=None):
        self.echo, self.sync = echo, sync
        self.phases = []          # (label, ms) of the top-level prove()/verify() phases
        self._stack = []

    def _now(self):
        if self.sync:
            self.sync()            # drain the device stream so a phase is charged its own kernels
        return time.perf_counter()

    def start(self, message=None, prefix=''):
        state = {'t0': self._now(), 'last': None, 'prefix': prefix, 'top': not self._stack}
        state['last'] = state['t0']
        self._stack.append(state)
        if message and self.echo:
            print(prefix + message)

        def log(msg):
            t = self._now()
            ms = (t - state['last']) * 1000
            state['last'] = t
            if state['top']:
                self.phases.append((msg, ms))
            else:
                self.phases.append(('  ' + msg, ms))
            if self.echo:
                print(f"{state['prefix']}{msg} in {ms:.1f} ms")
        log.state = state
        return log

    def sub(self, message=None):
        prefix = (self._stack[-1]['prefix'] if self._stack else '') + '  '
        return self.start(message, prefix)

    def done(self, log, message=None):
        state = log.state
        if state in self._stack:
            self._stack.remove(state)
        if message:
            ms = (self._now() - state['t0']) * 1000
            self.phases.append((message, ms))
            if self.echo:
                print(f"{state['prefix']}{message} in {ms:.1f} ms")


class NoopLogger:
    phases = []

    def start(self, message=None, prefix=''):
        return lambda msg: None

    def sub(self, message=None):
        return lambda msg: None

    def done(self, log, message=None):
        pass
