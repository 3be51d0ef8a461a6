// P/Invoke declarations for libdraco_mi355x.so (include/draco_mi355x.h).
// This file is what a draco-sharp maintainer adds next to src/Draco/IO/DracoDecoder.cs; it cannot be
// compiled in the build image (no .NET SDK), the ctypes binding in draco-sharp_amd/native.py is the
// executable twin used by the tests.  Blittable structs only, no callbacks, library-owned memory.
using System;
using System.Runtime.InteropServices;

namespace Draco.IO.Gpu;

internal enum DsaStatus : int
{
    Ok = 0,
    InvalidData = 1,      // -> InvalidDataException
    NotImplemented = 2,   // -> NotImplementedException
    InvalidArgument = 3,  // -> ArgumentException
    Device = 4,           // -> InvalidOperationException(dsa_last_error)
    OutOfMemory = 5       // -> OutOfMemoryException
}

[StructLayout(LayoutKind.Sequential)]
internal struct DsaMeshInfo
{
    public int Status;
    public int Detail;
    public byte MajorVersion, MinorVersion, EncoderType, EncoderMethod;
    public ushort Flags;
    public ushort DecodePath;   // 0 wave-per-mesh kernels, 1 general path, 2 general path at the second attempt
    public uint NumFaces;
    public uint NumPoints;
    public uint NumAttributes;
    public ulong DrcBytes;
}

[StructLayout(LayoutKind.Sequential)]
internal unsafe struct DsaAttributeInfo
{
    public int AttributeType;
    public int DataType;
    public int NumComponents;
    public int Normalized;
    public uint UniqueId;
    public uint NumEntries;
    public uint ByteStride;
    public int DecoderType;
    public int PredictionMethod;
    public int PredictionTransform;
    public int QuantizationBits;
    public float Range;
    public fixed float MinValues[4];
}

[StructLayout(LayoutKind.Sequential)]
internal unsafe struct DsaMeshOutput       // byte offsets of a mesh's arrays inside the batch's output block (dsa_batch_download)
{
    public uint Block;        // 0: the batch's block, 1: the block of the meshes decoded a second time (general path)
    public uint Flags;            // DSA_OUTPUT_FACES_U16 = 1 (compact download)
    public ulong Faces;
    public fixed ulong Values[16];
    public fixed ulong PointMap[16];
}

[StructLayout(LayoutKind.Sequential)]
internal struct DsaEncodeOptions
{
    public int PositionBits, TexcoordBits, NormalBits;
    public int SingleConnectivity;
    public int SymbolScheme;        // -1 auto, 0 tagged, 1 raw
    public int CompressionLevel;    // 10 - Config.Speed
    public int PositionPrediction, TexcoordPrediction;
}

[StructLayout(LayoutKind.Sequential)]
internal unsafe struct DsaMeshInput
{
    public uint NumVertices, NumFaces;
    public float* Positions;
    public uint* Faces;
    public float* Normals;
    public float* Texcoords;
    public byte* Generic;           // ABI 4: num_vertices * GenericComponents bytes or null
    public uint GenericComponents;
    public uint Reserved;
}

internal static unsafe partial class NativeMethods
{
    private const string Lib = "draco_mi355x";

    [DllImport(Lib)] internal static extern int dsa_abi_version();
    [DllImport(Lib)] internal static extern int dsa_device_count();
    [DllImport(Lib)] internal static extern DsaStatus dsa_context_create(int device, IntPtr stream, out IntPtr ctx);
    [DllImport(Lib)] internal static extern void dsa_context_destroy(IntPtr ctx);
    [DllImport(Lib)] internal static extern IntPtr dsa_last_error(IntPtr ctx);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_create(IntPtr ctx, uint n, byte** streams, nuint* lengths, out IntPtr batch);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_create_packed(IntPtr ctx, uint n, byte* blob, ulong* offsets, out IntPtr batch);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_decode(IntPtr batch);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_wait(IntPtr batch);
    [DllImport(Lib)] internal static extern void dsa_batch_free(IntPtr batch);
    [DllImport(Lib)] internal static extern uint dsa_batch_size(IntPtr batch);
    [DllImport(Lib)] internal static extern ulong dsa_batch_algorithmic_bytes(IntPtr batch);
    [DllImport(Lib)] internal static extern ulong dsa_batch_arena_bytes(IntPtr batch);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_mesh_info(IntPtr batch, uint mesh, out DsaMeshInfo info);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_attribute_info(IntPtr batch, uint mesh, uint attribute, out DsaAttributeInfo info);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_copy_faces(IntPtr batch, uint mesh, int* dst);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_copy_attribute_values(IntPtr batch, uint mesh, uint attribute, void* dst);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_copy_point_map(IntPtr batch, uint mesh, uint attribute, uint* dst);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_copy_portable_values(IntPtr batch, uint mesh, uint attribute, int* dst);
    [DllImport(Lib)] internal static extern IntPtr dsa_batch_device_faces(IntPtr batch, uint mesh);
    [DllImport(Lib)] internal static extern IntPtr dsa_batch_device_attribute_values(IntPtr batch, uint mesh, uint attribute);
    [DllImport(Lib)] internal static extern IntPtr dsa_batch_device_point_map(IntPtr batch, uint mesh, uint attribute);
    // whole-batch copy-out: one transfer of every output array into a pinned mirror (or caller memory), beside the next batch's kernels
    [DllImport(Lib)] internal static extern ulong dsa_batch_output_bytes(IntPtr batch);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_download(IntPtr batch, void* dst, nuint dstBytes);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_download_compact(IntPtr batch, void* dst, nuint dstBytes);
    [DllImport(Lib)] internal static extern ulong dsa_batch_compact_bytes(IntPtr batch);
    [DllImport(Lib)] internal static extern IntPtr dsa_batch_host_output(IntPtr batch, uint block);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_output_layout(IntPtr batch, uint mesh, out DsaMeshOutput layout);
    [DllImport(Lib)] internal static extern IntPtr dsa_host_alloc(nuint bytes);
    [DllImport(Lib)] internal static extern void dsa_host_free(IntPtr p);
    [DllImport(Lib)] internal static extern DsaStatus dsa_host_register(void* p, nuint bytes);
    [DllImport(Lib)] internal static extern DsaStatus dsa_host_unregister(void* p);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_copy_metadata(IntPtr batch, uint mesh, byte* dst, nuint dstBytes, out nuint length);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_copy_debug(IntPtr batch, uint mesh, int what, void* dst, nuint dstBytes, out nuint written);
    [DllImport(Lib)] internal static extern DsaStatus dsa_context_set_profiling(IntPtr ctx, int enabled);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_stage_times(IntPtr batch, float* ms, IntPtr* names);
    [DllImport(Lib)] internal static extern DsaStatus dsa_batch_kernel_times(IntPtr batch, float* ms, IntPtr* names, uint capacity, out uint count);
    [DllImport(Lib)] internal static extern DsaStatus dsa_context_trim(IntPtr ctx);
    [DllImport(Lib)] internal static extern IntPtr dsa_context_schedule_note(IntPtr ctx);

    // encode direction (DracoEncoder.Encode, src/Draco/IO/DracoEncoder.cs:22-41)
    [DllImport(Lib)] internal static extern void dsa_encode_default_options(out DsaEncodeOptions options);
    [DllImport(Lib)] internal static extern DsaStatus dsa_encode_batch(IntPtr ctx, uint n, DsaMeshInput* meshes, in DsaEncodeOptions options, out IntPtr encoded);
    [DllImport(Lib)] internal static extern uint dsa_encoded_size(IntPtr encoded);
    [DllImport(Lib)] internal static extern DsaStatus dsa_encoded_stream(IntPtr encoded, uint mesh, out byte* bytes, out nuint length);
    [DllImport(Lib)] internal static extern void dsa_encoded_free(IntPtr encoded);

    // multi-GPU submit for a single-process host: one context + worker thread per listed device inside the library
    [DllImport(Lib)] internal static extern DsaStatus dsa_pool_create(int* devices, uint numDevices, uint chunkMeshes, out IntPtr pool);
    [DllImport(Lib)] internal static extern void dsa_pool_destroy(IntPtr pool);
    [DllImport(Lib)] internal static extern uint dsa_pool_size(IntPtr pool);
    [DllImport(Lib)] internal static extern IntPtr dsa_pool_last_error(IntPtr pool);
    [DllImport(Lib)] internal static extern DsaStatus dsa_pool_decode(IntPtr pool, uint n, byte** streams, nuint* lengths, out IntPtr job);
    [DllImport(Lib)] internal static extern DsaStatus dsa_pool_job_locate(IntPtr job, uint stream, out IntPtr batch, out uint mesh, out uint worker);
    [DllImport(Lib)] internal static extern uint dsa_pool_job_chunks(IntPtr job);
    [DllImport(Lib)] internal static extern void dsa_pool_job_free(IntPtr job);
    [DllImport(Lib)] internal static extern uint dsa_pool_plan(uint n, nuint* lengths, uint chunkMeshes, uint* order, uint* chunkBegin);

    internal static void Check(DsaStatus status, IntPtr ctx, string what)
    {
        if (status == DsaStatus.Ok) return;
        var msg = ctx == IntPtr.Zero ? what : $"{what}: {Marshal.PtrToStringAnsi(dsa_last_error(ctx))}";
        throw status switch
        {
            DsaStatus.InvalidData => new System.IO.InvalidDataException(msg),
            DsaStatus.NotImplemented => new NotImplementedException(msg),
            DsaStatus.InvalidArgument => new ArgumentException(msg),
            DsaStatus.OutOfMemory => new OutOfMemoryException(msg),
            _ => new InvalidOperationException(msg),
        };
    }
}
