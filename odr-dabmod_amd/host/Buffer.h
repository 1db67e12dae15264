// Buffer.h -- host byte buffer with the interface of ODR-DabMod's Buffer
// (reference src/Buffer.h:51-93): 32-byte aligned storage that only
// reallocates when it grows and keeps its contents when it does.
// Written from scratch for the drop-in adapters; same member names and
// semantics so that a stage written against the reference compiles against it.
#pragma once

#include <complex>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

typedef std::complex<float> complexf;

class Buffer {
public:
    using sptr = std::shared_ptr<Buffer>;

    Buffer(size_t len = 0, const void *data = nullptr);
    Buffer(const Buffer &other);
    Buffer(Buffer &&other) noexcept;
    Buffer(const std::vector<uint8_t> &vec);
    ~Buffer();

    void swap(Buffer &other) noexcept;

    // Resize; reallocates (32-byte aligned, contents kept) only when growing.
    void setLength(size_t len);
    // Replace the contents.
    void setData(const void *data, size_t len);
    Buffer &operator=(const Buffer &other);
    Buffer &operator=(Buffer &&other) noexcept;
    Buffer &operator=(const std::vector<uint8_t> &buf);

    uint8_t operator[](size_t i) const;

    // Concatenate.
    void appendData(const void *data, size_t len);
    Buffer &operator+=(const Buffer &other);

    size_t getLength() const { return m_len; }
    void *getData() const { return m_data; }

private:
    size_t m_len = 0;
    size_t m_capacity = 0;
    void *m_data = nullptr;
};

void swap(Buffer &a, Buffer &b) noexcept;
