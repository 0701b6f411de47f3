// rk_javaser.hpp -- a reader for the Java Object Serialization Stream Protocol (what ObjectOutputStream writes), just enough of
// it to open a RAPPAS `.union` database without a JVM (src/main_v2/SessionNext_v2.java:109-207).  C++ twin of
// rappas_amd/javaser.py; the two are compared with each other in tests/test_host_cpp.py.
//
// Generic by construction: the stream describes every class it contains (name, serialVersionUID, flags, field list, super
// class), so objects of classes this file has never heard of -- the Swing JTree machinery tree.PhyloTree drags in, fastutil's
// maps -- are parsed from their own descriptors: default field data class by class (super class first), then, for classes with
// a writeObject method (SC_WRITE_METHOD), the "annotation" records up to TC_ENDBLOCKDATA.  Nothing is instantiated.  Grammar:
// Java Object Serialization Specification, chapter 6; handles are assigned in the order the specification gives (newHandle).
//
// HOST-SIDE INGEST ONLY; nothing here touches the placement path.  PARITY UNPINNED: no JVM exists
// in this environment, so this reader has only ever seen streams assembled by tests/javaser_writer.py from the same
// specification, never a file a JVM wrote.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rkjs {

enum : uint8_t {
    TC_NULL = 0x70, TC_REFERENCE, TC_CLASSDESC, TC_OBJECT, TC_STRING, TC_ARRAY, TC_CLASS, TC_BLOCKDATA, TC_ENDBLOCKDATA, TC_RESET,
    TC_BLOCKDATALONG, TC_EXCEPTION, TC_LONGSTRING, TC_PROXYCLASSDESC, TC_ENUM
};
enum : uint8_t { SC_WRITE_METHOD = 0x01, SC_SERIALIZABLE = 0x02, SC_EXTERNALIZABLE = 0x04, SC_BLOCK_DATA = 0x08 };
constexpr int32_t BASE_WIRE_HANDLE = 0x7E0000;

struct Error : std::runtime_error {
    Error(const std::string &msg, size_t offset) : std::runtime_error(msg + " (stream offset " + std::to_string(offset) + ")") {}
};

struct ClassDesc {
    std::string name;
    int64_t uid = 0;
    uint8_t flags = 0;
    struct Field { char type; std::string name; };
    std::vector<Field> fields;
    std::shared_ptr<ClassDesc> super;
    std::vector<const ClassDesc *> hierarchy() const {  // super class first, as class data is laid out in the stream
        std::vector<const ClassDesc *> chain;  // (finite: Reader::classdesc refuses a chain that comes back to itself)
        for (const ClassDesc *d = this; d; d = d->super.get()) chain.push_back(d);
        return std::vector<const ClassDesc *>(chain.rbegin(), chain.rend());
    }
};

struct Node;
using P = std::shared_ptr<Node>;  // nullptr = Java null

struct Node {
    enum Kind { PRIM, STRING, OBJECT, ARRAY, CLASSDESC, ENUM, BLOCK } kind;
    char ptype = 0;  // PRIM: B C D F I J S Z; ARRAY of primitives: the element type
    int64_t i = 0;   // PRIM integral value (C as its UTF-16 code unit, Z as 0/1); ARRAY of primitives: the element count
    double f = 0;    // PRIM D / F
    std::string s;   // STRING text, BLOCK bytes, ENUM constant name; ARRAY of primitives: the elements as the stream holds them (big-endian)
    std::shared_ptr<ClassDesc> desc;
    std::vector<std::pair<std::string, std::map<std::string, P>>> fields;  // OBJECT: class name -> {field name: value}
    std::vector<std::pair<std::string, std::vector<P>>> annotations;       // OBJECT: class name -> what its writeObject wrote after the fields
    std::vector<P> elems;                                                  // ARRAY of objects (primitive arrays stay raw in s: no node per element)
    explicit Node(Kind k) : kind(k) {}

    const std::string &classname() const { return desc->name; }
    static int prim_width(char t) { return t == 'B' || t == 'Z' ? 1 : t == 'C' || t == 'S' ? 2 : t == 'I' || t == 'F' ? 4 : t == 'J' || t == 'D' ? 8 : 0; }
    // element e of a primitive array as an integer (B S I J signed, C Z unsigned; F / D: their bit pattern)
    int64_t prim_at(size_t e) const {
        const int w = prim_width(ptype);
        if (kind != ARRAY || w == 0 || e >= (size_t)i) throw std::runtime_error("javaser: not an element of a primitive array");
        uint64_t v = 0;
        for (int b = 0; b < w; b++) v = v << 8 | (uint8_t)s[e * (size_t)w + (size_t)b];
        if (ptype == 'B') return (int8_t)v;
        if (ptype == 'S') return (int16_t)v;
        if (ptype == 'I') return (int32_t)v;
        return (int64_t)v;
    }
    // value of `field` looked up from the most derived class upwards; found = false when no class has it
    P get(const std::string &field, bool *found = nullptr) const {
        for (size_t c = fields.size(); c-- > 0;) {
            auto it = fields[c].second.find(field);
            if (it != fields[c].second.end()) {
                if (found) *found = true;
                return it->second;
            }
        }
        if (found) *found = false;
        return nullptr;
    }
    const std::vector<P> *annotation(const std::string &cls) const {
        for (const auto &a : annotations)
            if (a.first == cls) return &a.second;
        return nullptr;
    }
    std::string block(const std::string &cls) const {  // the block-data bytes of one class's annotations, concatenated
        std::string out;
        if (const auto *a = annotation(cls))
            for (const P &x : *a)
                if (x && x->kind == BLOCK) out += x->s;
        return out;
    }
    std::vector<P> objects(const std::string &cls) const {  // the same annotations without the block data (nulls kept)
        std::vector<P> out;
        if (const auto *a = annotation(cls))
            for (const P &x : *a)
                if (!x || x->kind != BLOCK) out.push_back(x);
        return out;
    }
};

struct Record {
    bool is_block;
    P value;  // BLOCK node, or the object (nullptr for a null written with writeObject)
};

class Reader {
  public:
    Reader(const uint8_t *data, size_t n) : d_(data), n_(n) {
        if (u2() != 0xACED || u2() != 5) throw Error("not a Java serialization stream (bad magic / version)", 0);
    }
    // top-level records: block data for primitive writes, objects for writeObject calls
    std::vector<Record> contents() {
        std::vector<Record> out;
        contents([&](size_t, Record &&r) { out.push_back(std::move(r)); return true; });
        return out;
    }
    // The same, record by record: `take` gets record number i and says whether it keeps it.  A record it does not keep is let go of
    // at once, with every object that was read for it (a `.union` file's alignment, extended tree and AR tree are of no use to a
    // placement); a later reference to one of those objects reads as a DROPPED placeholder.
    void contents(const std::function<bool(size_t, Record &&)> &take) {
        size_t i = 0;
        while (p_ < n_) {
            const uint8_t tc = d_[p_];
            if (tc == TC_BLOCKDATA || tc == TC_BLOCKDATALONG) { take(i++, Record{true, blockdata()}); continue; }
            if (tc == TC_RESET) { p_++; handles_.clear(); continue; }
            const size_t h0 = handles_.size();
            if (!take(i++, Record{false, content()})) drop_handles(h0);
        }
    }
    // Objects a class's writeObject wrote behind its fields (its "annotations") are handed to `sink` one by one instead of being kept
    // with the object -- the entries of a fastutil map with 10^7 rows, turned into CSR rows as they are read -- and let go of at once.
    void stream_annotations_of(const std::string &class_name, std::function<void(const P &)> sink) {
        sink_class_ = class_name;
        sink_ = std::move(sink);
    }

  private:
    const uint8_t *d_;
    size_t n_, p_ = 0;
    struct Handle { P node; std::shared_ptr<ClassDesc> desc; bool dropped = false; };
    std::vector<Handle> handles_;
    std::string sink_class_;
    std::function<void(const P &)> sink_;
    // objects read since handle h0 are let go of (class descriptors stay: later objects of the same classes refer to them)
    void drop_handles(size_t h0) {
        for (size_t h = h0; h < handles_.size(); h++)
            if (!handles_[h].desc) { handles_[h].node.reset(); handles_[h].dropped = true; }
    }
    // Nesting of content() / classdesc() calls.  ObjectOutputStream itself recurses once per nested object and a JVM's default
    // stack gives out after a few thousand levels, so no stream a JVM wrote comes near this; a crafted one stops here, not in a
    // stack overflow (each level costs three or four frames of this reader).
    static constexpr int MAX_DEPTH = 3000;
    int depth_ = 0;
    struct Nest {
        Reader &r;
        Nest(Reader &rd, size_t at) : r(rd) {
            if (++r.depth_ > MAX_DEPTH) { r.depth_--; throw Error("objects nested more than " + std::to_string(MAX_DEPTH) + " deep", at); }
        }
        ~Nest() { r.depth_--; }
    };

    const uint8_t *take(size_t n) {
        if (n > n_ - p_) throw Error("truncated stream", p_);
        const uint8_t *b = d_ + p_;
        p_ += n;
        return b;
    }
    uint8_t u1() { return *take(1); }
    uint16_t u2() { const uint8_t *b = take(2); return (uint16_t)(b[0] << 8 | b[1]); }
    int32_t i4() { const uint8_t *b = take(4); return (int32_t)((uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3]); }
    int64_t i8() { const uint64_t hi = (uint32_t)i4(), lo = (uint32_t)i4(); return (int64_t)(hi << 32 | lo); }
    std::string utf(bool lng = false) {
        const int64_t n = lng ? i8() : u2();
        if (n < 0) throw Error("negative string length", p_);
        const uint8_t *b = take((size_t)n);
        return std::string((const char *)b, (size_t)n);  // modified UTF-8 kept as is (names and labels here are ASCII)
    }
    size_t new_handle(P node, std::shared_ptr<ClassDesc> desc = nullptr) {
        handles_.push_back({std::move(node), std::move(desc)});
        return handles_.size() - 1;
    }
    P blockdata() {
        const uint8_t tc = u1();
        const int64_t n = tc == TC_BLOCKDATA ? u1() : i4();
        if (n < 0) throw Error("negative block length", p_);
        auto b = std::make_shared<Node>(Node::BLOCK);
        const uint8_t *raw = take((size_t)n);
        b->s.assign((const char *)raw, (size_t)n);
        return b;
    }
    P prim(char t) {
        auto v = std::make_shared<Node>(Node::PRIM);
        v->ptype = t;
        switch (t) {
        case 'B': v->i = (int8_t)u1(); break;
        case 'Z': v->i = u1() != 0; break;
        case 'C': v->i = u2(); break;
        case 'S': v->i = (int16_t)u2(); break;
        case 'I': v->i = i4(); break;
        case 'J': v->i = i8(); break;
        case 'F': { const uint32_t b = (uint32_t)i4(); float x; memcpy(&x, &b, 4); v->f = x; break; }
        case 'D': { const uint64_t b = (uint64_t)i8(); double x; memcpy(&x, &b, 8); v->f = x; break; }
        default: throw Error(std::string("unknown primitive type code '") + t + "'", p_);
        }
        return v;
    }
    static bool is_prim(char t) { return strchr("BCDFIJSZ", t) != nullptr && t != 0; }

    // one `object` production
    P content() {
        const size_t at = p_;
        const Nest nest(*this, at);
        const uint8_t tc = u1();
        switch (tc) {
        case TC_NULL: return nullptr;
        case TC_REFERENCE: {
            const int64_t h = (int64_t)i4() - BASE_WIRE_HANDLE;
            if (h < 0 || h >= (int64_t)handles_.size()) throw Error("back reference to unknown handle " + std::to_string(h), at);
            const Handle &hd = handles_[(size_t)h];
            if (hd.dropped) {  // an object of a record (or a map entry) the caller let go of: nothing that is kept may need it
                auto d = std::make_shared<Node>(Node::PRIM);
                d->ptype = 'X';
                d->s = "<dropped>";
                return d;
            }
            if (hd.desc) {  // a class descriptor used as a value
                auto c = std::make_shared<Node>(Node::CLASSDESC);
                c->desc = hd.desc;
                return c;
            }
            return hd.node;
        }
        case TC_STRING: case TC_LONGSTRING: {
            auto s = std::make_shared<Node>(Node::STRING);
            new_handle(s);
            s->s = utf(tc == TC_LONGSTRING);
            return s;
        }
        case TC_CLASSDESC: case TC_PROXYCLASSDESC: {
            p_ = at;
            auto c = std::make_shared<Node>(Node::CLASSDESC);
            c->desc = classdesc();
            return c;
        }
        case TC_CLASS: {
            auto c = std::make_shared<Node>(Node::CLASSDESC);
            c->desc = classdesc();
            new_handle(c);
            return c;
        }
        case TC_ENUM: {
            auto e = std::make_shared<Node>(Node::ENUM);
            e->desc = classdesc();
            new_handle(e);
            const P name = content();
            if (name && name->kind == Node::STRING) e->s = name->s;
            return e;
        }
        case TC_ARRAY: {
            auto a = std::make_shared<Node>(Node::ARRAY);
            a->desc = classdesc();
            if (!a->desc) throw Error("array without a class descriptor", at);
            new_handle(a);
            const int32_t n = i4();
            if (n < 0) throw Error("negative array length", at);
            const char t = a->desc->name.size() > 1 ? a->desc->name[1] : '?';
            if ((size_t)n > n_ - p_) throw Error("array longer than what is left of the stream", at);  // (every element takes >= 1 byte)
            if (is_prim(t)) {  // kept as the stream's bytes (byte[]: the k-mer keys; char[][]: the alignment), read with prim_at()
                const size_t bytes = (size_t)n * (size_t)Node::prim_width(t);
                const uint8_t *raw = take(bytes);
                a->s.assign((const char *)raw, bytes);
                a->ptype = t;
                a->i = n;
            } else {
                for (int32_t e = 0; e < n; e++) a->elems.push_back(content());
            }
            return a;
        }
        case TC_OBJECT: {
            auto o = std::make_shared<Node>(Node::OBJECT);
            o->desc = classdesc();
            if (!o->desc) throw Error("object without a class descriptor", at);
            new_handle(o);
            classdata(*o);
            return o;
        }
        case TC_EXCEPTION: throw Error("the stream records an exception thrown while it was written", at);
        case TC_BLOCKDATA: case TC_BLOCKDATALONG: case TC_ENDBLOCKDATA:
            throw Error("block data record where an object is expected", at);
        default: throw Error("unknown type code " + std::to_string(tc), at);
        }
    }

    // a descriptor's handle exists before its super class is read, so a crafted stream can name the descriptor itself (or one
    // whose chain leads back to it) as its own super class: refuse it, every walk up the chain relies on it ending
    void set_super(const std::shared_ptr<ClassDesc> &d, std::shared_ptr<ClassDesc> sup, size_t at) {
        for (const ClassDesc *c = sup.get(); c; c = c->super.get())
            if (c == d.get()) throw Error("class descriptor " + d->name + " is its own super class", at);
        d->super = std::move(sup);
    }

    std::shared_ptr<ClassDesc> classdesc() {
        const size_t at = p_;
        const Nest nest(*this, at);
        const uint8_t tc = u1();
        if (tc == TC_NULL) return nullptr;
        if (tc == TC_REFERENCE) {
            const int64_t h = (int64_t)i4() - BASE_WIRE_HANDLE;
            if (h < 0 || h >= (int64_t)handles_.size() || !handles_[(size_t)h].desc)
                throw Error("class descriptor reference does not name a class descriptor", at);
            return handles_[(size_t)h].desc;
        }
        if (tc == TC_PROXYCLASSDESC) {
            auto d = std::make_shared<ClassDesc>();
            d->name = "<proxy>";
            d->flags = SC_SERIALIZABLE;
            new_handle(nullptr, d);
            const int32_t n = i4();
            for (int32_t e = 0; e < n; e++) utf();
            annotations();
            set_super(d, classdesc(), at);
            return d;
        }
        if (tc != TC_CLASSDESC) throw Error("type code " + std::to_string(tc) + " where a class descriptor is expected", at);
        auto d = std::make_shared<ClassDesc>();
        d->name = utf();
        d->uid = i8();
        new_handle(nullptr, d);
        d->flags = u1();
        const uint16_t nf = u2();
        for (uint16_t e = 0; e < nf; e++) {
            const char t = (char)u1();
            std::string fname = utf();
            if (t == '[' || t == 'L') content();  // class name of an object / array field: a String object
            d->fields.push_back({t, std::move(fname)});
        }
        annotations();  // classAnnotation (annotateClass writes nothing by default)
        set_super(d, classdesc(), at);
        return d;
    }

    std::vector<P> annotations(bool to_sink = false) {
        std::vector<P> out;
        while (true) {
            if (p_ >= n_) throw Error("truncated stream inside an annotation", p_);
            const uint8_t tc = d_[p_];
            if (tc == TC_ENDBLOCKDATA) { p_++; return out; }
            if (tc == TC_BLOCKDATA || tc == TC_BLOCKDATALONG) out.push_back(blockdata());
            else if (tc == TC_RESET) p_++;
            else if (to_sink) {
                const size_t h0 = handles_.size();
                const P x = content();
                sink_(x);
                drop_handles(h0);
            } else out.push_back(content());
        }
    }

    void classdata(Node &obj) {
        for (const ClassDesc *d : obj.desc->hierarchy()) {
            if (d->flags & SC_SERIALIZABLE) {
                std::map<std::string, P> vals;
                for (const auto &f : d->fields) vals[f.name] = is_prim(f.type) ? prim(f.type) : content();
                obj.fields.emplace_back(d->name, std::move(vals));
                if (d->flags & SC_WRITE_METHOD) obj.annotations.emplace_back(d->name, annotations(sink_ && d->name == sink_class_));
            } else if (d->flags & SC_EXTERNALIZABLE) {
                if (!(d->flags & SC_BLOCK_DATA)) throw Error(d->name + ": Externalizable data of stream protocol 1 cannot be delimited", p_);
                obj.annotations.emplace_back(d->name, annotations());
            } else {
                throw Error(d->name + ": class descriptor is neither Serializable nor Externalizable", p_);
            }
        }
    }
};

inline std::vector<Record> parse(const std::string &data) { return Reader((const uint8_t *)data.data(), data.size()).contents(); }

// ---- java.util containers by their documented serial forms ----
// java.util.HashMap / LinkedHashMap: writeObject = defaultWriteObject, then block data {int buckets, int size}, then key, value
// objects alternately (java.util.HashMap.writeObject / internalWriteEntries)
inline std::vector<std::pair<P, P>> hashmap_items(const Node &obj) {
    const std::vector<P> objs = obj.objects("java.util.HashMap");
    std::vector<std::pair<P, P>> out;
    for (size_t e = 0; e + 1 < objs.size(); e += 2) out.emplace_back(objs[e], objs[e + 1]);
    return out;
}
// java.lang.Integer / Character ... -> the integral value (field `value` of the wrapper class)
inline int64_t boxed_int(const P &v) {
    if (!v) throw std::runtime_error("union: null where a boxed number is expected");
    if (v->kind == Node::PRIM) return v->i;
    const P x = v->get("value");
    if (!x || x->kind != Node::PRIM) throw std::runtime_error("union: " + (v->desc ? v->desc->name : std::string("value")) + " is not a boxed number");
    return x->i;
}
// java.util.ArrayList: defaultWriteObject (size), block data {int capacity}, then the elements; java.util.Vector: default fields
// (elementData array, elementCount)
inline std::vector<P> list_items(const Node &obj) {
    bool is_vector = obj.classname() == "java.util.Vector";
    for (const auto &f : obj.fields) is_vector = is_vector || f.first == "java.util.Vector";
    if (is_vector) {
        const P n = obj.get("elementCount"), data = obj.get("elementData");
        std::vector<P> out;
        if (data)
            for (size_t e = 0; e < data->elems.size() && (int64_t)e < (n ? n->i : 0); e++) out.push_back(data->elems[e]);
        return out;
    }
    return obj.objects("java.util.ArrayList");
}

}  // namespace rkjs
