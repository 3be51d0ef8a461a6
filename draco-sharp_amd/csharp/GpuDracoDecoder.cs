// Drop-in GPU path behind DracoDecoder.Decode (src/Draco/IO/DracoDecoder.cs:8-42): same inputs, same
// Draco / Mesh / PointAttribute results, same exception types; the connectivity and attribute decoding
// that the reference runs in-process (ConnectivityDecoder.DecodeConnectivity / DecodeAttributes) happen
// in libdraco_mi355x.so on an MI355X.  DecodeBatch is the call that makes the GPU worthwhile: the streams
// of one call are independent meshes decoded concurrently.
using System;
using System.Collections.Generic;
using System.IO;
using Draco.IO.Attributes;
using Draco.IO.Enums;
using Draco.IO.Metadata;

namespace Draco.IO.Gpu;

public sealed unsafe class GpuDracoDecoder : IDisposable
{
    private IntPtr _ctx;

    public GpuDracoDecoder(int device = 0)
    {
        NativeMethods.Check(NativeMethods.dsa_context_create(device, IntPtr.Zero, out _ctx), IntPtr.Zero, $"no usable GPU {device}");
    }

    public Draco Decode(string path)
    {
        return DecodeBatch([File.ReadAllBytes(path)])[0];
    }

    public Draco Decode(Stream stream)
    {
        using var ms = new MemoryStream();
        stream.CopyTo(ms);
        stream.Dispose();   // the reference disposes the caller's reader (DecoderBuffer.cs:186-189)
        return DecodeBatch([ms.ToArray()])[0];
    }

    public Draco Decode(BinaryReader binaryReader)
    {
        return Decode(binaryReader.BaseStream);
    }

    /// <summary>Batches in flight: submits a batch (parse, pinned staging, upload on the copy stream, kernels, download) and returns at
    /// once; Collect waits for that batch alone.  With two or three submitted ahead the upload of the next, the kernels of the
    /// current and the download of the previous one overlap (bench.py end_to_end: 23 k 64k-triangle meshes/s on one MI355X).</summary>
    public IntPtr Submit(IReadOnlyList<byte[]> streams)
    {
        var handles = new System.Runtime.InteropServices.GCHandle[streams.Count];
        var ptrs = new byte*[streams.Count];
        var lens = new nuint[streams.Count];
        IntPtr batch = IntPtr.Zero;
        try
        {
            for (int i = 0; i < streams.Count; ++i)
            {
                handles[i] = System.Runtime.InteropServices.GCHandle.Alloc(streams[i], System.Runtime.InteropServices.GCHandleType.Pinned);
                ptrs[i] = (byte*)handles[i].AddrOfPinnedObject();
                lens[i] = (nuint)streams[i].Length;
            }
            fixed (byte** p = ptrs)
            fixed (nuint* l = lens)
                NativeMethods.Check(NativeMethods.dsa_batch_create(_ctx, (uint)streams.Count, p, l, out batch), _ctx, "dsa_batch_create");   // the streams are staged before this returns
            NativeMethods.Check(NativeMethods.dsa_batch_decode(batch), _ctx, "dsa_batch_decode");
            // compact: uint16 faces where they fit, one point map per distinct map -- a third less on the link; the dsa_batch_copy_* calls
            // of Materialize widen from the host copy
            NativeMethods.Check(NativeMethods.dsa_batch_download_compact(batch, null, 0), _ctx, "dsa_batch_download_compact");
            return batch;
        }
        catch { if (batch != IntPtr.Zero) NativeMethods.dsa_batch_free(batch); throw; }
        finally { foreach (var h in handles) if (h.IsAllocated) h.Free(); }
    }

    public Draco[] Collect(IntPtr batch)
    {
        try
        {
            NativeMethods.Check(NativeMethods.dsa_batch_wait(batch), _ctx, "dsa_batch_wait");
            var results = new Draco[NativeMethods.dsa_batch_size(batch)];
            for (uint i = 0; i < results.Length; ++i) results[i] = Materialize(batch, i);
            return results;
        }
        finally { NativeMethods.dsa_batch_free(batch); }
    }

    /// <summary>Decodes independent .drc streams in one GPU batch.  A bad stream throws when its result is requested
    /// (InvalidDataException / NotImplementedException, as the reference would); it never poisons the others.</summary>
    public Draco[] DecodeBatch(IReadOnlyList<byte[]> streams)
    {
        var handles = new System.Runtime.InteropServices.GCHandle[streams.Count];
        var ptrs = new byte*[streams.Count];
        var lens = new nuint[streams.Count];
        IntPtr batch = IntPtr.Zero;
        try
        {
            for (int i = 0; i < streams.Count; ++i)
            {
                handles[i] = System.Runtime.InteropServices.GCHandle.Alloc(streams[i], System.Runtime.InteropServices.GCHandleType.Pinned);
                ptrs[i] = (byte*)handles[i].AddrOfPinnedObject();
                lens[i] = (nuint)streams[i].Length;
            }
            fixed (byte** p = ptrs)
            fixed (nuint* l = lens)
            {
                NativeMethods.Check(NativeMethods.dsa_batch_create(_ctx, (uint)streams.Count, p, l, out batch), _ctx, "dsa_batch_create");
            }
            NativeMethods.Check(NativeMethods.dsa_batch_decode(batch), _ctx, "dsa_batch_decode");
            // every output array of the batch in ONE device -> host transfer into a pinned mirror, queued behind the kernels; the
            // per-array copies of Materialize below are then served from that host copy (dsa_batch_copy_* after a download); the compact
            // form of the transfer: faces as uint16 where a mesh has at most 65 536 points, one point map per distinct map
            NativeMethods.Check(NativeMethods.dsa_batch_download_compact(batch, null, 0), _ctx, "dsa_batch_download_compact");
            NativeMethods.Check(NativeMethods.dsa_batch_wait(batch), _ctx, "dsa_batch_wait");
            var results = new Draco[streams.Count];
            for (uint i = 0; i < streams.Count; ++i) results[i] = Materialize(batch, i);
            return results;
        }
        finally
        {
            if (batch != IntPtr.Zero) NativeMethods.dsa_batch_free(batch);
            foreach (var h in handles) if (h.IsAllocated) h.Free();
        }
    }

    private Draco Materialize(IntPtr batch, uint mesh) => Materialize(_ctx, batch, mesh);

    internal static Draco Materialize(IntPtr _ctx, IntPtr batch, uint mesh)
    {
        NativeMethods.Check(NativeMethods.dsa_batch_mesh_info(batch, mesh, out var info), _ctx, "dsa_batch_mesh_info");
        NativeMethods.Check((DsaStatus)info.Status, IntPtr.Zero, $"stream {mesh}: decode failed (site {info.Detail})");
        // EncodedGeometryType.PointCloud (Constants.cs) -> PointCloud, TriangularMesh -> Mesh (DracoDecoder.cs:66-99)
        PointCloud.PointCloud result;
        if (info.EncoderType == 0) result = new PointCloud.PointCloud();
        else
        {
            var meshResult = new Mesh.Mesh();
            var faces = new int[info.NumFaces * 3];
            fixed (int* f = faces) NativeMethods.Check(NativeMethods.dsa_batch_copy_faces(batch, mesh, f), _ctx, "dsa_batch_copy_faces");
            meshResult.SetNumFaces((int)info.NumFaces);
            for (uint f = 0; f < info.NumFaces; ++f) meshResult.SetFace(f, [faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]]);
            result = meshResult;
        }
        result.PointsCount = (int)info.NumPoints;
        for (uint a = 0; a < info.NumAttributes; ++a)
        {
            NativeMethods.Check(NativeMethods.dsa_batch_attribute_info(batch, mesh, a, out var ai), _ctx, "dsa_batch_attribute_info");
            var geometry = new GeometryAttribute(
                attributeType: (GeometryAttributeType)ai.AttributeType, buffer: null, numComponents: (byte)ai.NumComponents,
                dataType: (DataType)ai.DataType, normalized: ai.Normalized != 0, byteStride: ai.ByteStride, byteOffset: 0)
            { UniqueId = ai.UniqueId };
            var attribute = new PointAttribute(geometry);
            attribute.Reset((int)ai.NumEntries);
            var values = new byte[(long)ai.NumEntries * ai.ByteStride];
            fixed (byte* v = values) NativeMethods.Check(NativeMethods.dsa_batch_copy_attribute_values(batch, mesh, a, v), _ctx, "dsa_batch_copy_attribute_values");
            attribute.Buffer!.Update(values);
            var map = new uint[info.NumPoints];
            fixed (uint* m = map) NativeMethods.Check(NativeMethods.dsa_batch_copy_point_map(batch, mesh, a, m), _ctx, "dsa_batch_copy_point_map");
            attribute.SetExplicitMapping((int)info.NumPoints);           // MeshTraversalSequencer.cs:33-50
            for (uint p = 0; p < info.NumPoints; ++p) attribute.SetPointMapEntry(p, map[p]);
            result.AddAttribute(attribute);
        }
        return new Draco
        {
            Header = new DracoHeader(info.MajorVersion, info.MinorVersion, info.EncoderType, info.EncoderMethod, info.Flags),
            Metadata = (info.Flags & 0x8000) != 0 ? ReadMetadata(_ctx, batch, mesh) : null,   // DracoDecoder.cs:23-28
            ConnectedData = result,
            Attributes = result.Attributes
        };
    }

    // The metadata block byte for byte from the native side (it only skips it), parsed into the reference's
    // DracoMetadata / MetadataElement here.  Layout: Metadata/MetadataDecoder.cs:5-49, with value sizes as varints
    // (what the bitstream writes; the managed decoder reads a single byte there).
    private static DracoMetadata ReadMetadata(IntPtr _ctx, IntPtr batch, uint mesh)
    {
        NativeMethods.Check(NativeMethods.dsa_batch_copy_metadata(batch, mesh, null, 0, out nuint length), _ctx, "dsa_batch_copy_metadata");
        var block = new byte[(int)length];
        fixed (byte* p = block) NativeMethods.Check(NativeMethods.dsa_batch_copy_metadata(batch, mesh, p, length, out _), _ctx, "dsa_batch_copy_metadata");
        int pos = 0;
        byte U8() => pos < block.Length ? block[pos++] : throw new InvalidDataException("metadata block truncated");
        ulong Varint()
        {
            ulong r = 0;
            for (int shift = 0; shift < 64; shift += 7) { byte b = U8(); r |= (ulong)(b & 0x7F) << shift; if ((b & 0x80) == 0) return r; }
            throw new InvalidDataException("metadata varint too long");
        }
        sbyte[] Take(ulong k)
        {
            if (k > (ulong)(block.Length - pos)) throw new InvalidDataException("metadata block truncated");
            var r = new sbyte[(int)k];
            Buffer.BlockCopy(block, pos, r, 0, (int)k);
            pos += (int)k;
            return r;
        }
        MetadataElement Element(int depth)
        {
            if (depth > 15) throw new InvalidDataException("metadata nesting too deep");
            ulong n = Varint();
            var keys = new List<sbyte[]>(); var values = new List<sbyte[]>();
            for (ulong i = 0; i < n; ++i) { keys.Add(Take(U8())); values.Add(Take(Varint())); }
            ulong ns = Varint();
            var subKeys = new List<sbyte[]>(); var subs = new List<MetadataElement>();
            for (ulong i = 0; i < ns; ++i) { subKeys.Add(Take(U8())); subs.Add(Element(depth + 1)); }
            return new MetadataElement { Keys = keys.ToArray(), Values = values.ToArray(), SubMetadataKeys = subKeys.ToArray(), SubMetadata = subs.ToArray() };
        }
        var attributes = new List<MetadataElement>();
        ulong count = Varint();
        for (ulong i = 0; i < count; ++i) { uint id = (uint)Varint(); var e = Element(0); e.Id = id; attributes.Add(e); }
        return new DracoMetadata { Attributes = attributes, File = Element(0) };
    }

    public void Dispose()
    {
        if (_ctx != IntPtr.Zero) { NativeMethods.dsa_context_destroy(_ctx); _ctx = IntPtr.Zero; }
    }
}

/// <summary>All GPUs of the node behind one managed object: the library owns one context and one worker thread per
/// listed device and hands the streams of a call out longest first through an atomic queue (dsa_pool_*; meshes are
/// independent, DracoDecoder.cs:19-42, so there is no collective).  A device may be listed more than once.</summary>
public sealed unsafe class GpuDracoDecoderPool : IDisposable
{
    private IntPtr _pool;

    public GpuDracoDecoderPool(IReadOnlyList<int> devices, uint chunkMeshes = 256)
    {
        var d = new int[devices.Count];
        for (int i = 0; i < d.Length; ++i) d[i] = devices[i];
        fixed (int* p = d) NativeMethods.Check(NativeMethods.dsa_pool_create(p, (uint)d.Length, chunkMeshes, out _pool), IntPtr.Zero, "no usable GPU in the device list");
    }

    public int Workers => (int)NativeMethods.dsa_pool_size(_pool);

    /// <summary>Decodes independent .drc streams on every GPU of the pool.  Element i of the result is stream i's Draco
    /// object, or null with errors[i] set to what the reference would have thrown for that stream.</summary>
    public Draco?[] DecodeBatch(IReadOnlyList<byte[]> streams, out Exception?[] errors)
    {
        var handles = new System.Runtime.InteropServices.GCHandle[streams.Count];
        var ptrs = new byte*[streams.Count];
        var lens = new nuint[streams.Count];
        IntPtr job = IntPtr.Zero;
        try
        {
            for (int i = 0; i < streams.Count; ++i)
            {
                handles[i] = System.Runtime.InteropServices.GCHandle.Alloc(streams[i], System.Runtime.InteropServices.GCHandleType.Pinned);
                ptrs[i] = (byte*)handles[i].AddrOfPinnedObject();
                lens[i] = (nuint)streams[i].Length;
            }
            DsaStatus st;
            fixed (byte** p = ptrs)
            fixed (nuint* l = lens)
            {
                st = NativeMethods.dsa_pool_decode(_pool, (uint)streams.Count, p, l, out job);
            }
            if (st != DsaStatus.Ok)
                NativeMethods.Check(st, IntPtr.Zero, "dsa_pool_decode: " + System.Runtime.InteropServices.Marshal.PtrToStringAnsi(NativeMethods.dsa_pool_last_error(_pool)));
            var results = new Draco?[streams.Count];
            errors = new Exception?[streams.Count];
            for (uint i = 0; i < streams.Count; ++i)
            {
                NativeMethods.Check(NativeMethods.dsa_pool_job_locate(job, i, out var batch, out var mesh, out _), IntPtr.Zero, "dsa_pool_job_locate");
                try { results[i] = GpuDracoDecoder.Materialize(IntPtr.Zero, batch, mesh); }
                catch (Exception e) when (e is InvalidDataException or NotImplementedException) { errors[i] = e; }
            }
            return results;
        }
        finally
        {
            if (job != IntPtr.Zero) NativeMethods.dsa_pool_job_free(job);
            foreach (var h in handles) if (h.IsAllocated) h.Free();
        }
    }

    public void Dispose()
    {
        if (_pool != IntPtr.Zero) { NativeMethods.dsa_pool_destroy(_pool); _pool = IntPtr.Zero; }
    }
}
