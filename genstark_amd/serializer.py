"""Mirror of lib/Serializer.ts:17-150 — proof wire format."""
from . import utils
from .errors import StarkError


class Serializer:
    def __init__(self, config, hashDigestSize, strict=False):
        self.strict = strict          # True: parseProof also rejects bytes after the last field (the reference ignores them)
        self.fieldElementSize = config.field.elementSize
        self.tRegisterCount = config.traceRegisterCount
        self.sRegisterCount = config.secretInputCount
        self.hashDigestSize = hashDigestSize

    def _valueCount(self):
        return self.tRegisterCount + self.sRegisterCount

    def serializeProof(self, proof):  # :35-79
        size = utils.sizeOf(proof, self.fieldElementSize, self.hashDigestSize)
        buffer = bytearray(size['total'])
        ds = self.hashDigestSize
        buffer[0:ds] = proof['evRoot']
        offset = ds
        offset = utils.writeMerkleProof(buffer, offset, proof['evProof'], self._valueCount() * self.fieldElementSize)
        ldLeafSize = self.fieldElementSize * 4
        ld = proof['ldProof']
        buffer[offset:offset + ds] = ld['lcRoot']
        offset += ds
        offset = utils.writeMerkleProof(buffer, offset, ld['lcProof'], ldLeafSize)
        buffer[offset] = len(ld['components'])
        offset += 1
        for component in ld['components']:
            buffer[offset:offset + ds] = component['columnRoot']
            offset += ds
            offset = utils.writeMerkleProof(buffer, offset, component['columnProof'], ldLeafSize)
            offset = utils.writeMerkleProof(buffer, offset, component['polyProof'], ldLeafSize)
        remainder = ld['remainder']
        buffer[offset] = 0 if len(remainder) == 256 else len(remainder)  # zero means 256 (:58-63)
        offset += 1
        for value in remainder:
            offset = utils.writeBigInt(value, buffer, offset, self.fieldElementSize)
        buffer[offset] = len(proof['iShapes'])
        offset += 1
        for shape in proof['iShapes']:
            buffer[offset] = len(shape)
            offset += 1
            for level in shape:
                buffer[offset:offset + 4] = int(level).to_bytes(4, 'little')
                offset += 4
        assert offset == len(buffer)
        return bytes(buffer)

    def parseProof(self, buffer):  # :83-144
        ds, es = self.hashDigestSize, self.fieldElementSize
        need = utils.need                                  # every read is bounds-checked: a truncated proof is a StarkError
        need(buffer, 0, ds)
        evRoot = bytes(buffer[0:ds])
        evProof, offset = utils.readMerkleProof(buffer, ds, self._valueCount() * es, ds)
        need(buffer, offset, ds)
        lcRoot = bytes(buffer[offset:offset + ds])
        offset += ds
        lcProof, offset = utils.readMerkleProof(buffer, offset, es * 4, ds)
        need(buffer, offset, 1)
        componentCount = buffer[offset]
        offset += 1
        components = []
        for _ in range(componentCount):
            need(buffer, offset, ds)
            columnRoot = bytes(buffer[offset:offset + ds])
            offset += ds
            columnProof, offset = utils.readMerkleProof(buffer, offset, es * 4, ds)
            polyProof, offset = utils.readMerkleProof(buffer, offset, es * 4, ds)
            components.append({'columnRoot': columnRoot, 'columnProof': columnProof, 'polyProof': polyProof})
        need(buffer, offset, 1)
        remainderLength = buffer[offset] or utils.MAX_ARRAY_LENGTH
        offset += 1
        remainder = []
        for _ in range(remainderLength):
            remainder.append(utils.readBigInt(buffer, offset, es))
            offset += es
        need(buffer, offset, 1)
        inputCount = buffer[offset]
        offset += 1
        inputShapes = []
        for _ in range(inputCount):
            need(buffer, offset, 1)
            rank = buffer[offset]
            offset += 1
            need(buffer, offset, 4 * rank)
            shape = []
            for _ in range(rank):
                shape.append(int.from_bytes(buffer[offset:offset + 4], 'little'))
                offset += 4
            inputShapes.append(shape)
        if self.strict and offset != len(buffer):      # the reference's parseProof ignores trailing bytes (lib/Serializer.ts:81-144): opt-in only
            raise StarkError('malformed proof: bytes left over after the last field')
        return {'evRoot': evRoot, 'evProof': evProof,
                'ldProof': {'lcRoot': lcRoot, 'lcProof': lcProof, 'components': components, 'remainder': remainder},
                'iShapes': inputShapes}
