"""A reader for the Java Object Serialization Stream Protocol (what `ObjectOutputStream` writes), just enough of it to open
a RAPPAS `.union` database without a JVM (src/main_v2/SessionNext_v2.java:109-207).

Generic by construction: the stream describes every class it contains (name, serialVersionUID, flags, field list, super
class), so objects of classes this module has never heard of -- the Swing `JTree` machinery `tree.PhyloTree` drags in, fastutil's
maps -- are parsed from their own descriptors: default field data class by class (super class first), then, for classes with a
`writeObject` method (SC_WRITE_METHOD), the "annotation" records up to TC_ENDBLOCKDATA.  Nothing is instantiated; an object
comes back as a `JavaObject` with `.fields[class name][field name]` and `.annotations[class name]` (a list of `bytes` block-data
records and nested objects).  Grammar: Java Object Serialization Specification, chapter 6 ("Object Serialization Stream
Protocol"); handles are assigned in the order the specification gives (`newHandle`).

HOST-SIDE INGEST ONLY; nothing here touches the placement path.  PARITY UNPINNED: no JVM exists
in this environment, so this reader has only ever seen streams assembled by tests/javaser_writer.py from the same
specification, never a file a JVM wrote.
"""
import struct

STREAM_MAGIC, STREAM_VERSION = 0xACED, 5
TC_NULL, TC_REFERENCE, TC_CLASSDESC, TC_OBJECT, TC_STRING, TC_ARRAY, TC_CLASS = 0x70, 0x71, 0x72, 0x73, 0x74, 0x75, 0x76
TC_BLOCKDATA, TC_ENDBLOCKDATA, TC_RESET, TC_BLOCKDATALONG, TC_EXCEPTION, TC_LONGSTRING = 0x77, 0x78, 0x79, 0x7A, 0x7B, 0x7C
TC_PROXYCLASSDESC, TC_ENUM = 0x7D, 0x7E
BASE_WIRE_HANDLE = 0x7E0000
SC_WRITE_METHOD, SC_SERIALIZABLE, SC_EXTERNALIZABLE, SC_BLOCK_DATA, SC_ENUM = 0x01, 0x02, 0x04, 0x08, 0x10
PRIM = {"B": ">b", "C": ">H", "D": ">d", "F": ">f", "I": ">i", "J": ">q", "S": ">h", "Z": ">?"}


class JavaSerializationError(ValueError):
    def __init__(self, msg, offset):
        super().__init__(f"{msg} (stream offset {offset})")
        self.offset = offset


class JavaClassDesc:
    def __init__(self, name, uid, flags, fields, superdesc):
        self.name, self.uid, self.flags, self.fields, self.superdesc = name, uid, flags, fields, superdesc

    def hierarchy(self):
        """super class first, as class data is laid out in the stream"""
        chain, d = [], self
        while d is not None:
            chain.append(d)
            d = d.superdesc
        return chain[::-1]

    def __repr__(self):
        return f"<classdesc {self.name}>"


class JavaObject:
    def __init__(self, desc):
        self.desc = desc
        self.fields = {}       # class name -> {field name: value}
        self.annotations = {}  # class name -> [bytes | object, ...]  (what that class's writeObject wrote after its fields)

    @property
    def classname(self):
        return self.desc.name

    def get(self, field, default=None):
        """value of `field` looked up from the most derived class upwards"""
        for d in self.desc.hierarchy()[::-1]:
            if field in self.fields.get(d.name, {}):
                return self.fields[d.name][field]
        return default

    def block(self, classname):
        """the block-data bytes of one class's annotations, concatenated"""
        return b"".join(x for x in self.annotations.get(classname, []) if isinstance(x, (bytes, bytearray)))

    def objects(self, classname):
        return [x for x in self.annotations.get(classname, []) if not isinstance(x, (bytes, bytearray))]

    def __repr__(self):
        return f"<{self.desc.name}>"


class JavaArray(list):
    def __init__(self, desc, values):
        super().__init__(values)
        self.desc = desc


class JavaEnum:
    def __init__(self, desc, name):
        self.desc, self.name = desc, name


class Reader:
    def __init__(self, data):
        self.d = memoryview(bytes(data))
        self.p = 0
        self.handles = []
        if self._u2() != STREAM_MAGIC or self._u2() != STREAM_VERSION:
            raise JavaSerializationError("not a Java serialization stream (bad magic / version)", 0)

    # ---- primitives ----
    def _take(self, n):
        if self.p + n > len(self.d):
            raise JavaSerializationError("truncated stream", self.p)
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def _u1(self):
        return self._take(1)[0]

    def _u2(self):
        return struct.unpack(">H", self._take(2))[0]

    def _i4(self):
        return struct.unpack(">i", self._take(4))[0]

    def _i8(self):
        return struct.unpack(">q", self._take(8))[0]

    def _utf(self, long=False):
        n = self._i8() if long else self._u2()
        raw = bytes(self._take(n))
        try:
            return raw.decode("utf-8")
        except UnicodeDecodeError:  # modified UTF-8 (embedded NUL as C0 80, surrogate pairs): keep the text readable
            return raw.replace(b"\xc0\x80", b"\x00").decode("utf-8", "replace")

    def _new_handle(self, obj):
        self.handles.append(obj)
        return len(self.handles) - 1

    # ---- grammar ----
    def contents(self):
        """top-level records: ('block', bytes) for primitive data, ('object', value) for writeObject calls"""
        while self.p < len(self.d):
            tc = self.d[self.p]
            if tc == TC_BLOCKDATA or tc == TC_BLOCKDATALONG:
                yield "block", self._blockdata()
            elif tc == TC_RESET:
                self.p += 1
                self.handles = []
            else:
                at = self.p
                try:
                    yield "object", self.content()
                except RecursionError:  # (rk_javaser.hpp stops at 3000 levels; here the interpreter's own limit does)
                    raise JavaSerializationError("objects nested too deep", at) from None

    def _blockdata(self):
        tc = self._u1()
        n = self._u1() if tc == TC_BLOCKDATA else self._i4()
        return bytes(self._take(n))

    MAX_DEPTH = 3000  # (rk_javaser.hpp's cap; a JVM's own stack gives out long before)

    def content(self):
        """one `object` production"""
        self.depth = getattr(self, "depth", 0) + 1
        try:
            if self.depth > self.MAX_DEPTH:
                raise JavaSerializationError(f"objects nested too deep (more than {self.MAX_DEPTH} levels)", self.p)
            return self._content()
        finally:
            self.depth -= 1

    def _content(self):
        at = self.p
        tc = self._u1()
        if tc == TC_NULL:
            return None
        if tc == TC_REFERENCE:
            h = self._i4() - BASE_WIRE_HANDLE
            if not 0 <= h < len(self.handles):
                raise JavaSerializationError(f"back reference to unknown handle {h}", at)
            return self.handles[h]
        if tc == TC_STRING or tc == TC_LONGSTRING:
            h = self._new_handle(None)
            s = self._utf(long=tc == TC_LONGSTRING)
            self.handles[h] = s
            return s
        if tc == TC_CLASSDESC or tc == TC_PROXYCLASSDESC:
            self.p = at
            return self._classdesc()
        if tc == TC_CLASS:
            d = self._classdesc()
            self._new_handle(d)
            return d
        if tc == TC_ENUM:
            d = self._classdesc()
            h = self._new_handle(None)
            e = JavaEnum(d, self.content())
            self.handles[h] = e
            return e
        if tc == TC_ARRAY:
            d = self._classdesc()
            h = self._new_handle(None)
            n = self._i4()
            t = d.name[1] if len(d.name) > 1 else "?"
            if t == "B":
                vals = JavaArray(d, [])
                vals.raw = bytes(self._take(n))  # byte[]: kept as bytes (k-mer keys)
            elif t in PRIM:
                size = struct.calcsize(PRIM[t])
                vals = JavaArray(d, struct.unpack(">" + PRIM[t][1] * n, self._take(n * size)))
            else:
                vals = JavaArray(d, [])
                self.handles[h] = vals
                for _ in range(n):
                    vals.append(self.content())
            self.handles[h] = vals
            return vals
        if tc == TC_OBJECT:
            d = self._classdesc()
            if d is None:
                raise JavaSerializationError("object without a class descriptor", at)
            obj = JavaObject(d)
            self._new_handle(obj)
            self._classdata(obj)
            return obj
        if tc == TC_EXCEPTION:
            raise JavaSerializationError("the stream records an exception thrown while it was written", at)
        if tc == TC_BLOCKDATA or tc == TC_BLOCKDATALONG or tc == TC_ENDBLOCKDATA:
            raise JavaSerializationError(f"block data record 0x{tc:02x} where an object is expected", at)
        raise JavaSerializationError(f"unknown type code 0x{tc:02x}", at)

    def _set_super(self, d, sup, at):
        """a descriptor's handle exists before its super class is read: refuse a chain that comes back to the descriptor"""
        c = sup
        while c is not None:
            if c is d:
                raise JavaSerializationError(f"class descriptor {d.name} is its own super class", at)
            c = c.superdesc
        d.superdesc = sup

    def _classdesc(self):
        at = self.p
        tc = self._u1()
        if tc == TC_NULL:
            return None
        if tc == TC_REFERENCE:
            h = self._i4() - BASE_WIRE_HANDLE
            if not 0 <= h < len(self.handles) or not isinstance(self.handles[h], JavaClassDesc):
                raise JavaSerializationError("class descriptor reference does not name a class descriptor", at)
            return self.handles[h]
        if tc == TC_PROXYCLASSDESC:
            d = JavaClassDesc("<proxy>", 0, SC_SERIALIZABLE, [], None)
            self._new_handle(d)
            d.interfaces = [self._utf() for _ in range(self._i4())]
            self._annotations()
            self._set_super(d, self._classdesc(), at)
            return d
        if tc != TC_CLASSDESC:
            raise JavaSerializationError(f"type code 0x{tc:02x} where a class descriptor is expected", at)
        name = self._utf()
        uid = self._i8()
        d = JavaClassDesc(name, uid, 0, [], None)
        self._new_handle(d)
        d.flags = self._u1()
        for _ in range(self._u2()):
            t = chr(self._u1())
            fname = self._utf()
            ftype = self.content() if t in "[L" else t  # class name of an object / array field: a String object
            d.fields.append((t, fname, ftype))
        self._annotations()  # classAnnotation (annotateClass writes nothing by default)
        self._set_super(d, self._classdesc(), at)
        return d

    def _annotations(self):
        out = []
        while True:
            tc = self.d[self.p] if self.p < len(self.d) else None
            if tc is None:
                raise JavaSerializationError("truncated stream inside an annotation", self.p)
            if tc == TC_ENDBLOCKDATA:
                self.p += 1
                return out
            if tc == TC_BLOCKDATA or tc == TC_BLOCKDATALONG:
                out.append(self._blockdata())
            elif tc == TC_RESET:
                self.p += 1
            else:
                out.append(self.content())

    def _classdata(self, obj):
        for d in obj.desc.hierarchy():
            if d.flags & SC_SERIALIZABLE:
                vals = {}
                for t, fname, _ in d.fields:
                    if t in PRIM:
                        vals[fname] = struct.unpack(PRIM[t], self._take(struct.calcsize(PRIM[t])))[0]
                    else:
                        vals[fname] = self.content()
                obj.fields[d.name] = vals
                if d.flags & SC_WRITE_METHOD:
                    obj.annotations[d.name] = self._annotations()
            elif d.flags & SC_EXTERNALIZABLE:
                if not d.flags & SC_BLOCK_DATA:
                    raise JavaSerializationError(f"{d.name}: Externalizable data of stream protocol 1 cannot be delimited", self.p)
                obj.annotations[d.name] = self._annotations()
            else:
                raise JavaSerializationError(f"{d.name}: class descriptor is neither Serializable nor Externalizable", self.p)


def parse(data):
    """list of top-level ('block', bytes) / ('object', value) records of a serialization stream"""
    return list(Reader(data).contents())


# ------------------------------------------------------------------------------------------------------------------
# java.util / fastutil containers by their documented serial forms
# ------------------------------------------------------------------------------------------------------------------
def hashmap_items(obj):
    """java.util.HashMap / LinkedHashMap: writeObject = defaultWriteObject, then block data {int buckets, int size}, then
    key, value objects alternately (java.util.HashMap.writeObject / internalWriteEntries)."""
    objs = obj.objects("java.util.HashMap")
    return list(zip(objs[0::2], objs[1::2]))


def boxed(v):
    """java.lang.Integer / Float / Character ... -> the Python value (field `value` of the wrapper class)"""
    return v.get("value") if isinstance(v, JavaObject) else v


def arraylist_items(obj):
    """java.util.ArrayList: defaultWriteObject (size), block data {int capacity}, then the elements.
    java.util.Vector: default fields (elementData array, elementCount)."""
    if obj.classname == "java.util.Vector" or "java.util.Vector" in obj.fields:
        n = obj.get("elementCount", 0)
        return list(obj.get("elementData") or [])[:n]
    return obj.objects("java.util.ArrayList")
